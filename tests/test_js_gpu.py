"""GPU: the Node.js host (js/vpt) -> N-API addon -> C-ABI -> HIP kernels, checked inside node against the JS ray-march."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_host_on_gpu():
    addon = os.path.join(ROOT, "js", "addon", "vpt_native.node")
    assert os.path.exists(addon), "build the addon first: make -C js/addon (or __graft_entry__.build())"
    res = subprocess.run([NODE, os.path.join(ROOT, "js", "test", "test_gpu.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode()
    assert res.returncode == 0, out
    assert "js gpu ok" in out


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_node_host_without_gpu():
    """addon loads, matrix recipe bit-exact vs the gl-matrix fixture, PropertyBag / factory semantics"""
    addon = os.path.join(ROOT, "js", "addon", "vpt_native.node")
    if not os.path.exists(addon):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "js", "addon")])
    res = subprocess.run([NODE, os.path.join(ROOT, "js", "test", "test_host.js")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert res.returncode == 0, res.stdout.decode()
