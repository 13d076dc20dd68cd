"""plain_reading.py — a SECOND, independent CPU restatement of the four hot-path shaders, written straight from the GLSL
"as written".  TEST INFRASTRUCTURE ONLY (imported by tests/ — never by the product, bench.py's timed path or smoke()).

Why it exists: oracle/vpt_oracle.c is the *contract* the HIP kernels are bit-exact against, and the contract makes
instruction-level choices (software reciprocal / rsqrt, polynomial log / sin / cos, constant-first matrix products, fma
forms, a hoisted blur == 0 case).  A shared misreading of the shaders could not show up in a bit-exact comparison of the
two.  This file makes none of those choices:

  * every `/` is an IEEE division, sqrt is IEEE, log / sin / cos / atan / asin / pow are libm in float64 rounded to float32;
  * every +, -, * is one float32 operation in the order the GLSL expression is written (left to right, no fma,
    matrix * vector as the sum over columns m[0]*x + m[1]*y + m[2]*z + m[3]*w);
  * built-ins by their GLSL ES 3.00 definitions: mix(a,b,t) = a*(1-t) + b*t, mod(x,y) = x - y*floor(x/y),
    normalize(v) = v / length(v), distance = length(a - b), dot = x*x' + y*y' + z*z';
  * textures by the OpenGL ES 3.0 spec (section 3.8.9-3.8.11): texel centres at (k+1/2)/N, CLAMP_TO_EDGE on the texel index,
    LINEAR = weighted sum of the 2^d normalised texels with weights (1-a)(1-b)(1-c)..., R8 -> v/255, SRGB8_ALPHA8 ->
    rgb decoded by the sRGB EOTF before filtering, alpha linear;
  * nothing is hoisted: unprojectRand draws and uses its disk sample whatever `blur` is.

What it shares with the contract (inputs, not arithmetic): the per-pixel varyings of the "fixed-seed mode"  —
NDC x = fl((2i+1)/W) - 1, vPosition[0,1] = fl(x*0.5 + 0.5), unproject evaluated at the pixel — and the integer PCG hash,
whose bits the shader fixes (mixins/random/hash/pcg.glsl:3-7).

Vectorised over pixels with numpy; sizes of 64x64 on 32^3 volumes run in seconds.  Reference lines are cited per function.
"""
import numpy as np

F = np.float32
U = np.uint32


def f32(x):
    return np.asarray(x, dtype=np.float32)


def _libm(fn, *xs):
    """libm in float64, rounded once to float32"""
    with np.errstate(all="ignore"):
        return fn(*[np.asarray(x, np.float64) for x in xs]).astype(np.float32)


# ---- mixins/random/hash/pcg.glsl:3-7, squashlinear.glsl:7-9, distribution/uniformdivision.glsl:3-6 -------------------
def pcg(x):
    x = np.asarray(x, np.uint64)
    x = (x * np.uint64(747796405) + np.uint64(2891336453)) & np.uint64(0xFFFFFFFF)
    x = ((((x >> ((x >> np.uint64(28)) + np.uint64(4))) ^ x) * np.uint64(277803737))) & np.uint64(0xFFFFFFFF)
    return ((x >> np.uint64(22)) ^ x) & np.uint64(0xFFFFFFFF)


def hash3(x, y, z):
    m = np.uint64(0xFFFFFFFF)
    v = (np.uint64(19) * x.astype(np.uint64) + np.uint64(47) * y.astype(np.uint64) + np.uint64(101) * z.astype(np.uint64) + np.uint64(131)) & m
    return pcg(v)


class Rng:
    """`inout uint state` of the distribution mixins, one state per pixel; draws only advance the lanes in `mask`"""

    def __init__(self, state):
        self.state = np.asarray(state, np.uint64).copy()

    def uniform(self, mask=None):
        new = pcg(self.state)
        if mask is not None:
            new = np.where(mask, new, self.state)
        self.state = new
        # float(state) / float(~0u): uint -> float rounds to nearest, ~0u rounds to 2^32; IEEE division
        return new.astype(np.float32) / np.float32(4294967295.0)

    def exponential(self, rate, mask=None):              # distribution/exponential.glsl:3-5
        u = self.uniform(mask)
        return -_libm(np.log, u) / F(rate)

    def square(self, mask=None):                         # distribution/square.glsl:3-7
        x = self.uniform(mask)
        y = self.uniform(mask)
        return x, y

    def disk(self, mask=None):                           # distribution/disk.glsl:3-7 (TWOPI from constants.glsl)
        radius = np.sqrt(self.uniform(mask))
        angle = F(6.28318530718) * self.uniform(mask)
        return radius * _libm(np.cos, angle), radius * _libm(np.sin, angle)

    def sphere(self, mask=None):                         # distribution/sphere.glsl:4-10
        dx, dy = self.disk(mask)
        norm = dx * dx + dy * dy
        with np.errstate(invalid="ignore"):
            radius = F(2.0) * np.sqrt(F(1.0) - norm)
        z = F(1.0) - F(2.0) * norm
        return radius * dx, radius * dy, z


# ---- textures (OpenGL ES 3.0 §3.8) -----------------------------------------------------------------------------------
def _linear_axis(s, n):
    u = s * F(n) - F(0.5)
    with np.errstate(invalid="ignore"):
        i0 = np.floor(u)
    a = u - i0                                           # frac(u - 1/2) of the spec
    bad = ~np.isfinite(u)
    i0 = np.where(bad, 0, i0)
    i0 = np.clip(i0, -2, n + 1).astype(np.int64)
    i1 = i0 + 1
    return np.clip(i0, 0, n - 1), np.clip(i1, 0, n - 1), np.where(bad, F(0), a).astype(np.float32)


def _nearest_axis(s, n):
    with np.errstate(invalid="ignore"):
        i = np.floor(s * F(n))
    i = np.where(np.isfinite(i), i, 0)
    return np.clip(i, 0, n - 1).astype(np.int64)


class Scene:
    def __init__(self, volume_u8, filter="linear", tf_rgba8=None, env_rgba8=None):
        v = np.asarray(volume_u8)
        if v.ndim == 3:
            v = v[..., None]
        self.vol = v.astype(np.float32) / F(255.0)       # normalised texels [z][y][x][c]
        self.filter = filter
        if tf_rgba8 is None:                            # AbstractRenderer.js:31-44
            tf_rgba8 = np.array([[[255, 0, 0, 0], [255, 0, 0, 255]]], np.uint8)
        tf = np.asarray(tf_rgba8, np.uint8)
        c = tf[..., :3].astype(np.float64) / 255.0
        lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)       # sRGB EOTF (ES 3.0 §3.8.16)
        self.tf = np.concatenate([lin, tf[..., 3:4].astype(np.float64) / 255.0], axis=-1).astype(np.float32)   # [h][w][4]
        if env_rgba8 is None:                           # RenderingContext.js:90-101
            env_rgba8 = np.array([[[255, 255, 255, 255]]], np.uint8)
        self.env = (np.asarray(env_rgba8, np.uint8).astype(np.float32) / F(255.0))

    def volume_rg(self, px, py, pz):
        """texture(uVolume, position).rg"""
        nz, ny, nx, nc = self.vol.shape
        if self.filter == "nearest":
            x, y, z = _nearest_axis(px, nx), _nearest_axis(py, ny), _nearest_axis(pz, nz)
            t = self.vol[z, y, x]
        else:
            x0, x1, a = _linear_axis(px, nx)
            y0, y1, b = _linear_axis(py, ny)
            z0, z1, c = _linear_axis(pz, nz)
            one = F(1.0)
            t = 0
            for zz, wz in ((z0, one - c), (z1, c)):
                for yy, wy in ((y0, one - b), (y1, b)):
                    for xx, wx in ((x0, one - a), (x1, a)):
                        t = t + ((wx * wy) * wz)[..., None] * self.vol[zz, yy, xx]
            t = t.astype(np.float32)
        r = t[..., 0]
        g = t[..., 1] if nc > 1 else np.zeros_like(r)
        return r, g

    @staticmethod
    def _bilinear(tex, s, t):
        h, w, _ = tex.shape
        x0, x1, a = _linear_axis(s, w)
        y0, y1, b = _linear_axis(t, h)
        one = F(1.0)
        out = (((one - a) * (one - b))[..., None] * tex[y0, x0] + (a * (one - b))[..., None] * tex[y0, x1]
               + ((one - a) * b)[..., None] * tex[y1, x0] + (a * b)[..., None] * tex[y1, x1])
        return out.astype(np.float32)

    def color(self, px, py, pz):
        """sampleVolumeColor (MIPRenderer.glsl:45-49 = EAM :46-50, MCS :64-68, MCM :85-89)"""
        r, g = self.volume_rg(px, py, pz)
        return self._bilinear(self.tf, r, g)

    def environment(self, dx, dy, dz):
        """sampleEnvironmentMap (MCSRenderer.glsl:59-62, MCMRenderer.glsl:80-83); INVPI from constants.glsl"""
        s = _libm(np.arctan2, dx, -dz) * F(0.31830988618) * F(0.5) + F(0.5)
        t = (_libm(np.arcsin, -dy) * F(2.0)) * F(0.31830988618) * F(0.5) + F(0.5)
        return self._bilinear(self.env, s, t)


# ---- ray set-up ------------------------------------------------------------------------------------------------------
def pixel_grid(w, h):
    """the fixed-seed mode's varyings (shared INPUT definition): NDC and [0,1] positions of pixel centres"""
    i = np.arange(w, dtype=np.float32)
    j = np.arange(h, dtype=np.float32)
    ndc_x = (F(2.0) * i + F(1.0)) / F(w) - F(1.0)
    ndc_y = (F(2.0) * j + F(1.0)) / F(h) - F(1.0)
    X, Y = np.meshgrid(ndc_x, ndc_y)                    # row 0 = bottom row (GL)
    return X.astype(np.float32), Y.astype(np.float32)


def mat_vec(m, x, y, z, w):
    """inverseMvp * vec4: column-major, summed over columns left to right"""
    m = np.asarray(m, np.float32).reshape(16)
    return tuple(((m[k] * x + m[4 + k] * y) + m[8 + k] * z) + m[12 + k] * w for k in range(4))


def unproject(px, py, m):
    """mixins/unproject.glsl:3-10"""
    one = np.ones_like(px)
    fx, fy, fz, fw = mat_vec(m, px, py, -one, one)
    tx, ty, tz, tw = mat_vec(m, px, py, one, one)
    return (fx / fw, fy / fw, fz / fw), (tx / tw, ty / tw, tz / tw)


def intersect_cube(o, d):
    """mixins/intersectCube.glsl:3-11"""
    with np.errstate(all="ignore"):
        tmin = [(F(0.0) - o[k]) / d[k] for k in range(3)]
        tmax = [(F(1.0) - o[k]) / d[k] for k in range(3)]
        t1 = [np.fmin(tmin[k], tmax[k]) for k in range(3)]
        t2 = [np.fmax(tmin[k], tmax[k]) for k in range(3)]
        tnear = np.fmax(np.fmax(t1[0], t1[1]), t1[2])
        tfar = np.fmin(np.fmin(t2[0], t2[1]), t2[2])
    return tnear, tfar


def mix3(a, b, t):
    return tuple(a[k] * (F(1.0) - t) + b[k] * t for k in range(3))


def dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def length3(a):
    with np.errstate(invalid="ignore"):
        return np.sqrt(dot3(a, a))


def normalize3(a):
    with np.errstate(all="ignore"):
        n = length3(a)
        return tuple(a[k] / n for k in range(3))


def to_unorm8(x):
    with np.errstate(invalid="ignore"):
        return np.rint(np.clip(np.nan_to_num(x, nan=0.0), 0.0, 1.0) * F(255.0)).astype(np.uint8)


# ---- MIP: MIPRenderer.glsl:51-72 (generate), :105-109 (integrate) --------------------------------------------------------
def mip_generate(scene, w, h, m, step_size, offset):
    px, py = pixel_grid(w, h)
    rf, rt = unproject(px, py, m)
    d = tuple(rt[k] - rf[k] for k in range(3))
    tn, tf_ = intersect_cube(rf, d)
    tn = np.fmax(tn, F(0.0)); tf_ = np.fmax(tf_, F(0.0))
    hit = ~(tn >= tf_)
    frm = mix3(rf, rt, tn); to = mix3(rf, rt, tf_)
    t = np.zeros_like(px); val = np.zeros_like(px); off = np.full_like(px, F(offset))
    run = hit.copy()
    step = F(step_size)
    while run.any():
        pos = mix3(frm, to, off)
        a = scene.color(*pos)[..., 3]
        val = np.where(run, np.fmax(a, val), val)
        t = np.where(run, t + step, t)
        s = off + step
        off = np.where(run, s - np.floor(s), off)        # mod(x, 1.0)
        run = run & (t < F(1.0))
    return to_unorm8(np.where(hit, val, F(0.0)))


# ---- EAM: EAMRenderer.glsl:52-80 ------------------------------------------------------------------------------------------
def eam_generate(scene, w, h, m, step_size, offset, extinction):
    px, py = pixel_grid(w, h)
    rf, rt = unproject(px, py, m)
    d = tuple(rt[k] - rf[k] for k in range(3))
    tn, tf_ = intersect_cube(rf, d)
    tn = np.fmax(tn, F(0.0)); tf_ = np.fmax(tf_, F(0.0))
    hit = ~(tn >= tf_)
    frm = mix3(rf, rt, tn); to = mix3(rf, rt, tf_)
    step = F(step_size)
    ray_step = length3(tuple(frm[k] - to[k] for k in range(3))) * step
    t = np.full_like(px, step * F(offset))
    acc = np.zeros(px.shape + (4,), np.float32)
    run = hit & (t < F(1.0))
    while run.any():
        c = scene.color(*mix3(frm, to, t)).copy()
        c[..., 3] = c[..., 3] * (ray_step * F(extinction))
        c[..., :3] = c[..., :3] * c[..., 3:4]
        new = acc + (F(1.0) - acc[..., 3:4]) * c
        acc = np.where(run[..., None], new, acc).astype(np.float32)
        t = np.where(run, t + step, t)
        run = run & (t < F(1.0)) & (acc[..., 3] < F(0.99))
    big = acc[..., 3] > F(1.0)
    with np.errstate(all="ignore"):
        rgb = np.where(big[..., None], acc[..., :3] / acc[..., 3:4], acc[..., :3])
    out = np.zeros(px.shape + (4,), np.uint8)
    out[..., :3] = np.where(hit[..., None], to_unorm8(rgb), 0)
    out[..., 3] = 255
    return out


def eam_integrate(acc_u8, frame_u8, mix):
    """EAMRenderer.glsl:115-119 on RGBA8 buffers"""
    a = acc_u8.astype(np.float32) / F(255.0); f = frame_u8.astype(np.float32) / F(255.0)
    return to_unorm8(a * (F(1.0) - F(mix)) + f * F(mix))


# ---- MCS: MCSRenderer.glsl:70-137 -------------------------------------------------------------------------------------------
def _mcs_track(scene, rng, frm, to, ext, active, shadow, max_iters=100000):
    """sampleDistance (:70-87, shadow = False) / sampleTransmittance (:89-105, shadow = True)"""
    maxd = length3(tuple(frm[k] - to[k] for k in range(3)))
    dist = np.zeros_like(maxd); tr = np.ones_like(maxd)
    run = active.copy()
    n = 0
    while run.any() and n < max_iters:
        n += 1
        e = rng.exponential(ext, run)
        dist = np.where(run, dist + e, dist)
        with np.errstate(invalid="ignore"):
            esc = run & (dist > maxd)
        run = run & ~esc
        with np.errstate(all="ignore"):
            pos = mix3(frm, to, dist / maxd)
        a = scene.color(*pos)[..., 3]
        if shadow:
            tr = np.where(run, tr * (F(1.0) - a), tr)
        else:
            u = rng.uniform(run)
            run = run & ~(u < a)
    return dist, tr, maxd


def mcs_generate(scene, w, h, m, seed, extinction, light):
    px, py = pixel_grid(w, h)
    vpx = px * F(0.5) + F(0.5); vpy = py * F(0.5) + F(0.5)
    rf, rt = unproject(px, py, m)
    d = tuple(rt[k] - rf[k] for k in range(3))
    du = normalize3(d)
    tn, tf_ = intersect_cube(rf, d)
    tn = np.fmax(tn, F(0.0)); tf_ = np.fmax(tf_, F(0.0))
    out = scene.environment(*du).copy()
    hit = ~(tn >= tf_)
    frm = mix3(rf, rt, tn); to = mix3(rf, rt, tf_)
    state = hash3(vpx.view(np.uint32), vpy.view(np.uint32), np.full(px.shape, F(seed)).view(np.uint32))
    rng = Rng(state)
    dist, _, maxd = _mcs_track(scene, rng, frm, to, extinction, hit, False)
    with np.errstate(invalid="ignore"):
        scat = hit & ~(dist > maxd)
    with np.errstate(all="ignore"):
        x = mix3(frm, to, dist / maxd)
    L = tuple(np.full_like(px, F(light[k])) for k in range(3))
    _, tfar = intersect_cube(x, L)
    tfar = np.fmax(tfar, F(0.0))
    to2 = tuple(x[k] + L[k] * tfar for k in range(3))
    diffuse = scene.color(*x)
    lightc = scene.environment(*L)
    _, tr, _ = _mcs_track(scene, rng, x, to2, extinction, scat, True)
    col = (diffuse * lightc) * tr[..., None]
    return np.where(scat[..., None], col, out).astype(np.float32)


def mcs_integrate(acc, frame, inv_n):
    """MCSRenderer.glsl:173-177"""
    return (acc + (frame - acc) * F(inv_n)).astype(np.float32)


# ---- MCM: MCMRenderer.glsl:70-172, 259-275; mixins/unprojectRand.glsl:3-24 ----------------------------------------------------
def _unproject_rand(rng, px, py, m, inv_res, blur, mask):
    ox, oy = rng.disk(mask)
    ox = ox * F(blur); oy = oy * F(blur)
    sx, sy = rng.square(mask)
    ax = (sx * F(2.0) - F(1.0)) * F(inv_res[0]); ay = (sy * F(2.0) - F(1.0)) * F(inv_res[1])
    one = np.ones_like(px)
    fx, fy, fz, fw = mat_vec(m, px + ox, py + oy, -one, one)
    tx, ty, tz, tw = mat_vec(m, px + ax, py + ay, one, one)
    with np.errstate(all="ignore"):
        return (fx / fw, fy / fw, fz / fw), (tx / tw, ty / tw, tz / tw)


class McmState:
    def __init__(self, w, h):
        self.w, self.h = w, h
        self.pos = [np.zeros((h, w), np.float32) for _ in range(3)]
        self.dir = [np.zeros((h, w), np.float32) for _ in range(3)]
        self.tr = [np.ones((h, w), np.float32) for _ in range(3)]
        self.rad = [np.ones((h, w), np.float32) for _ in range(3)]
        self.bounces = np.zeros((h, w), np.uint32)
        self.samples = np.zeros((h, w), np.uint32)


def _reset_photon(st, rng, px, py, m, inv_res, blur, mask):
    """resetPhoton, MCMRenderer.glsl:70-78"""
    frm, to = _unproject_rand(rng, px, py, m, inv_res, blur, mask)
    d = normalize3(tuple(to[k] - frm[k] for k in range(3)))
    tn, _ = intersect_cube(frm, d)
    tn = np.fmax(tn, F(0.0))
    for k in range(3):
        st.dir[k] = np.where(mask, d[k], st.dir[k]).astype(np.float32)
        st.pos[k] = np.where(mask, frm[k] + tn * d[k], st.pos[k]).astype(np.float32)
        st.tr[k] = np.where(mask, F(1.0), st.tr[k]).astype(np.float32)
    st.bounces = np.where(mask, 0, st.bounces).astype(np.uint32)


def mcm_reset(w, h, m, seed, blur=0.0):
    """reset/fragment main(): MCMRenderer.glsl:259-275 — seeded from the NDC position"""
    px, py = pixel_grid(w, h)
    st = McmState(w, h)
    rng = Rng(hash3(px.view(np.uint32), py.view(np.uint32), np.full(px.shape, F(seed)).view(np.uint32)))
    _reset_photon(st, rng, px, py, m, (F(1.0) / F(w), F(1.0) / F(h)), blur, np.ones((h, w), bool))
    for k in range(3):
        st.rad[k][:] = 1.0
    st.samples[:] = 0
    return st


def mcm_integrate(scene, st, m, seed, extinction, anisotropy, max_bounces, steps, blur=0.0, trace=None):
    """integrate/fragment main(): MCMRenderer.glsl:116-172.  trace (optional list) receives per event the outcome code
    of every pixel: 0 null, 1 scattering, 2 absorption, 3 out of bounds."""
    w, h = st.w, st.h
    px, py = pixel_grid(w, h)
    inv_res = (F(1.0) / F(w), F(1.0) / F(h))
    mpx = px * F(0.5) + F(0.5); mpy = py * F(0.5) + F(0.5)
    rng = Rng(hash3(mpx.view(np.uint32), mpy.view(np.uint32), np.full(px.shape, F(seed)).view(np.uint32)))
    g = F(anisotropy)
    for _ in range(int(steps)):
        dist = rng.exponential(extinction)
        with np.errstate(all="ignore"):
            for k in range(3):
                st.pos[k] = (st.pos[k] + dist * st.dir[k]).astype(np.float32)
        vs = scene.color(st.pos[0], st.pos[1], st.pos[2])
        p_null = F(1.0) - vs[..., 3]
        mx = np.fmax(np.fmax(vs[..., 0], vs[..., 1]), vs[..., 2])
        p_scat = np.where(st.bounces >= U(max_bounces), F(0.0), vs[..., 3] * mx).astype(np.float32)
        p_abs = F(1.0) - p_null - p_scat
        wheel = rng.uniform()
        with np.errstate(invalid="ignore"):
            oob = ((st.pos[0] > 1) | (st.pos[1] > 1) | (st.pos[2] > 1) | (st.pos[0] < 0) | (st.pos[1] < 0) | (st.pos[2] < 0))
            absorb = ~oob & (wheel < p_abs)
            scatter = ~oob & ~absorb & (wheel < p_abs + p_scat)
        if trace is not None:
            trace.append(np.where(oob, 3, np.where(absorb, 2, np.where(scatter, 1, 0))).astype(np.uint8))
        # out of bounds / absorption: deposit, then resetPhoton
        env = scene.environment(st.dir[0], st.dir[1], st.dir[2])
        fin = oob | absorb
        st.samples = np.where(fin, st.samples + 1, st.samples).astype(np.uint32)
        with np.errstate(all="ignore"):
            n = st.samples.astype(np.float32)
            for k in range(3):
                radiance = np.where(oob, st.tr[k] * env[..., k], F(0.0)).astype(np.float32)
                st.rad[k] = np.where(fin, st.rad[k] + (radiance - st.rad[k]) / n, st.rad[k]).astype(np.float32)
        _reset_photon(st, rng, px, py, m, inv_res, blur, fin)
        # scattering: MCMRenderer.glsl:91-106
        if scatter.any():
            ux, uy, uz = rng.sphere(scatter)
            if abs(float(g)) < 1e-5:
                nd = (ux, uy, uz)
            else:
                g2 = g * g
                with np.errstate(all="ignore"):
                    c = (F(1.0) - g2) / (F(1.0) - g + F(2.0) * g * rng.uniform(scatter))
                    hgcos = (F(1.0) + g2 - c * c) / (F(2.0) * g)
                    ud = dot3((ux, uy, uz), st.dir)
                    circ = normalize3(tuple((ux, uy, uz)[k] - ud * st.dir[k] for k in range(3)))
                    sq = np.sqrt(F(1.0) - hgcos * hgcos)
                    nd = tuple(sq * circ[k] + hgcos * st.dir[k] for k in range(3))
            for k in range(3):
                st.tr[k] = np.where(scatter, st.tr[k] * vs[..., k], st.tr[k]).astype(np.float32)
                st.dir[k] = np.where(scatter, nd[k], st.dir[k]).astype(np.float32)
            st.bounces = np.where(scatter, st.bounces + 1, st.bounces).astype(np.uint32)
    return st
