# Convenience targets; the driver uses __graft_entry__.build() / smoke() and bench.py directly.
PY ?= python

build:            ## libvpt_hip.so (hipcc, gfx950), the CPU oracle, the N-API addon
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## CPU suite (no GPU needed)
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## parity suite on an MI355X
	$(PY) -m pytest tests -q -m gpu

bench:            ## one JSON line: samples/s, roofline, CPU baselines
	$(PY) bench.py

fixtures:         ## regenerate the golden fixtures (build container only: needs /root/reference and node)
	node tests/golden/make_mvp_fixture.js
	$(PY) tests/golden/make_pcg_kat.py
	$(PY) tests/golden/make_contract_fixture.py
	$(PY) tests/golden/make_tonemap_fixture.py
	$(PY) tests/golden/make_reader_fixture.py
	cd tests/golden && node --no-warnings --experimental-loader ./esm_loader.mjs run_reference_animator.mjs > circle_animator_r01.json
	cd tests/golden && node --no-warnings --experimental-loader ./esm_loader.mjs run_reference_orbit.mjs > orbit_animator_r01.json

clean:
	$(MAKE) -C vpt_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -f js/addon/vpt_native.node

.PHONY: build test test-gpu bench fixtures clean
