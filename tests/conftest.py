import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """the built libraries are git-ignored: on a checkout without them, build them in-tree (hipcc cross-compiles without a GPU)"""
    import subprocess
    lib = os.path.join(ROOT, "vpt_amd", "libvpt_hip.so")
    addon = os.path.join(ROOT, "js", "addon", "vpt_native.node")
    try:
        if not os.path.exists(lib):
            subprocess.run(["make", "-C", os.path.join(ROOT, "vpt_amd", "csrc")], check=False, stdout=subprocess.DEVNULL)
        if os.path.exists(lib) and not os.path.exists(addon):
            subprocess.run(["make", "-C", os.path.join(ROOT, "js", "addon")], check=False, stdout=subprocess.DEVNULL)
    except OSError:
        pass                                         # no make / hipcc: the tests that need the library fail loudly themselves


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import vpt_amd
    ctx = vpt_amd.Context(0)
    yield ctx
    ctx.destroy()


def default_matrix(aspect=1.0):
    from vpt_amd.scene import default_camera, Transform, Node, mvp_inverse_matrix
    return mvp_inverse_matrix(default_camera(aspect), Transform(Node()))


def orbit_camera(aspect, yaw=0.6, pitch=-0.35, dist=1.7):
    """camera orbiting the cube centre (non-axis-aligned rays: exercises all three slab axes)"""
    import math
    from vpt_amd.scene import Node, PerspectiveCamera, quat
    node = Node()
    qy = quat.setAxisAngle(quat.create(), [0, 1, 0], yaw)
    qx = quat.setAxisAngle(quat.create(), [1, 0, 0], pitch)
    q = quat.multiply(quat.create(), qy, qx)
    node.transform.localRotation = q
    # camera looks down -z in its own frame: position = R * (0,0,dist)
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    node.transform.localTranslation = [dist * sy * cp, -dist * sp, dist * cy * cp]
    cam = PerspectiveCamera(node)
    cam.aspect = aspect
    node.components.append(cam)
    return node
