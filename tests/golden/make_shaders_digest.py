#!/usr/bin/env python3
"""SURVEY section 8c pin (iii): a drift detector for the GLSL the oracle restates.  Runs the reference's own shader packer (bin/packer, plain
Node.js) IN PLACE on the reference's src/ tree — nothing is copied; the outputs go to a temporary directory — and records the SHA-256 and size
of the packed shaders.json / mixins.json plus the program names per renderer and tone mapper.  If the reference's shaders ever change, the
digests change, and every `file:line` citation in oracle/ and vpt_amd/csrc/ has to be re-read.

    python tests/golden/make_shaders_digest.py            -> tests/golden/shaders_digest.json   (build container only: needs /root/reference, node)
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

REFERENCE = os.environ.get("VPT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def digest():
    node = shutil.which("node")
    if node is None or not os.path.isdir(os.path.join(REFERENCE, "src", "glsl")):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        config = {"input": [{"input": os.path.join(REFERENCE, "src"), "output": "/", "filter": "\\.(glsl)$", "recursive": True, "parse": True}],
                  "transform": [],
                  "output": [{"input": "/glsl/shaders", "output": os.path.join(tmp, "shaders.json"), "mode": "json"},
                             {"input": "/glsl/mixins", "output": os.path.join(tmp, "mixins.json"), "mode": "json"}]}
        path = os.path.join(tmp, "packer.json")
        json.dump(config, open(path, "w"))
        res = subprocess.run([node, os.path.join(REFERENCE, "bin", "packer"), path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        if res.returncode != 0:
            raise RuntimeError("packer failed: " + res.stderr.decode()[-500:])
        out = {}
        for name in ("shaders.json", "mixins.json"):
            raw = open(os.path.join(tmp, name), "rb").read()
            out[name] = {"sha256": hashlib.sha256(raw).hexdigest(), "bytes": len(raw)}
        shaders = json.load(open(os.path.join(tmp, "shaders.json")))
        def names(node, depth):
            """the key tree down to `depth` levels (group / class / program / stage), leaves dropped"""
            if not isinstance(node, dict) or depth == 0:
                return None
            return {k: names(v, depth - 1) for k, v in sorted(node.items())}
        out["programs"] = names(shaders, 3)
        out["mixins"] = names(json.load(open(os.path.join(tmp, "mixins.json"))), 4)
    return out


if __name__ == "__main__":
    d = digest()
    if d is None:
        sys.exit("needs node and %s/src/glsl" % REFERENCE)
    d["_made_by"] = "tests/golden/make_shaders_digest.py: the reference's bin/packer run under node on /root/reference/src (GLSL parts only)"
    json.dump(d, open(os.path.join(HERE, "shaders_digest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in d.items() if k.endswith(".json")}))
