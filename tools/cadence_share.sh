mkdir -p gpurun_out/r04b; out=gpurun_out/r04b/cadence_share.txt; : > $out
for sh in "" "0,2,8" "1,4,8" "3,8,8"; do
  for pf in 0 16; do
    timeout -k 10 200 python3 tools/ab_mcm.py --tag "share[$sh]" ${sh:+--shard $sh} --play-fused $pf 2>/dev/null >> $out || exit 1
  done
done
cat $out
