// vpt_kernels_mcm.h — the MCM pass kernels (gfx950, wave64): the headline path; included by vpt_mcm.hip only.
#pragma once
#include "vpt_kernels.h"

// =============================================================================================
// MCM — MCMRenderer.glsl, mixins/Photon.glsl, mixins/unprojectRand.glsl
// =============================================================================================
struct Photon {
    f3 position, direction, transmittance, radiance;
    uint32_t bounces, samples;
};
// mixins/unprojectRand.glsl:3-24.  The near-plane point depends on the pixel only when blur == 0 (the reference
// always passes 0: MCMRenderer.js:93,157): the two disk uniforms are still drawn, their product with 0 is an exact
// zero, so `from` equals the un-jittered unprojection `from0` computed once per pixel.
// NOBLUR: the caller knows blur == 0 (k_mcm_miss: the tile classes are only used with it), the general branch is not compiled
template <bool NOBLUR = false>
VPT_DEV void unproject_rand(uint32_t &state, float px, float py, const PassArgs &a, f3 from0, f3 &from, f3 &to) {
    if (NOBLUR || a.blur == 0.0f) {
        random_uniform(state); random_uniform(state);
        from = from0;
    } else {
        f2 d = random_disk(state);
        from = dehomogenize(mat4_mul_point(a.mvp_inv, px + d.x * a.blur, py + d.y * a.blur, -1.0f));
    }
    float sx = random_uniform(state), sy = random_uniform(state);
    float ax = fmaf(sx, 2.0f, -1.0f) * a.inv_w;
    float ay = fmaf(sy, 2.0f, -1.0f) * a.inv_h;
    to = dehomogenize(mat4_mul_point(a.mvp_inv, px + ax, py + ay, 1.0f));
}
VPT_DEV f3 unproject_near(float px, float py, const PassArgs &a) {
    return dehomogenize(mat4_mul_point(a.mvp_inv, px + 0.0f, py + 0.0f, -1.0f));
}
// resetPhoton: MCMRenderer.glsl:70-78
// the photon's start on its ray: from + max(tnear, 0) * direction (MCMRenderer.glsl:75-77).  A function of (from, direction) alone:
// the kernels of cube-missing tiles (k_mcm_miss) recompute it from the stored direction instead of storing it.
VPT_DEV f3 photon_start(f3 from, f3 dir) {
    float tnear = vmax(intersect_cube_near(from, dir), 0.0f);
    return madd3(from, tnear, dir);
}
template <bool NOBLUR = false>
VPT_DEV void reset_photon(uint32_t &state, Photon &ph, float px, float py, const PassArgs &a, f3 from0) {
    f3 from, to;
    unproject_rand<NOBLUR>(state, px, py, a, from0, from, to);
    ph.direction = normalize3(sub3(to, from));
    ph.bounces = 0u;
    ph.position = photon_start(from, ph.direction);
    ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
}
// sampleHenyeyGreensteinAngleCosine: MCMRenderer.glsl:91-95
VPT_DEV float hg_cos(uint32_t &state, float g) {
    float g2 = g * g;
    float c = (1.0f - g2) * rcp_nr(fmaf(2.0f * g, random_uniform(state), 1.0f - g));
    return fmaf(-c, c, 1.0f + g2) * rcp_nr(2.0f * g);
}
// sampleHenyeyGreenstein: MCMRenderer.glsl:97-106
VPT_DEV f3 sample_hg(uint32_t &state, float g, f3 dir) {
    f3 u = random_sphere(state);
    if (fabsf(g) < 1e-5f) return u;
    float hgcos = hg_cos(state, g);
    float ud = dot3(u, dir);
    f3 c = { fmaf(-ud, dir.x, u.x), fmaf(-ud, dir.y, u.y), fmaf(-ud, dir.z, u.z) };
    c = normalize3(c);
    float s = sqrt_nr(fmaf(-hgcos, hgcos, 1.0f));
    return f3{ fmaf(s, c.x, hgcos * dir.x), fmaf(s, c.y, hgcos * dir.y), fmaf(s, c.z, hgcos * dir.z) };
}
// radiance += (rad - radiance) / float(samples)   (MCMRenderer.glsl:147-150,154-157), as * (1/n)
VPT_DEV void photon_deposit(Photon &ph, f3 rad) {
    ph.samples++;
    float inv_n = rcp_nr((float)ph.samples);
    ph.radiance.x += (rad.x - ph.radiance.x) * inv_n;
    ph.radiance.y += (rad.y - ph.radiance.y) * inv_n;
    ph.radiance.z += (rad.z - ph.radiance.z) * inv_n;
}

// reset/fragment main(): MCMRenderer.glsl:259-275 (seeded from the NDC position)
#ifdef VPT_MCM_PLAIN_KERNELS   // the two non-template kernels are compiled by one translation unit (vpt_mcm.hip)
__global__ void __launch_bounds__(VPT_BLOCK) k_mcm_reset(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.tile) return;
    Photon ph;
    if (p.valid) {
        float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
        uint32_t state = hash3(__float_as_uint(px), __float_as_uint(py), __float_as_uint(a.seed));
        reset_photon(state, ph, px, py, a, unproject_near(px, py, a));
    } else {
        ph.position = f3{ 0.0f, 0.0f, 0.0f };
        ph.direction = f3{ 0.0f, 0.0f, 1.0f };
    }
    ((f3 *)a.st0)[p.k] = ph.position;
    a.st1[p.k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, 0.0f);
    ((f3 *)a.st2)[p.k] = f3{ 1.0f, 1.0f, 1.0f };
    a.st3[p.k] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
}
#endif

// the `steps` delta-tracking events of one pixel on its persistent photon: MCMRenderer.glsl:128-166
// sampleVolumeColor of an event.  The reference samples BEFORE its bounds test (MCMRenderer.glsl:132-142), so the sample of an
// event that leaves the cube is taken at the clamped position (CLAMP_TO_EDGE) and then discarded.  It is executed here too, for
// every lane (the empty asm keeps the compiler from proving it dead on the out-of-bounds path): an out-of-cube position has at
// least one coordinate clamped onto the first or last voxel plane of its axis with weight 0 there, so its eight-tap footprint
// degenerates to four taps on a face of the volume — fetched from the boundary atlas (vpt_device.h, sample_volume_boundary:
// one aligned dword instead of two unaligned 8-byte gathers; bit-identical value).  At the benchmark camera 93 % of the events
// end outside the cube (80 % of the pixels never meet it), and the pass was bound by the texture path's gather rate.
// LINEAR one-channel byte volumes: the sample in two phases.  A wave whose lanes disagree (HIT tiles: some photons inside the cube,
// some outside) would run the two samplers one after the other, each waiting for its own load: mcm_sample_issue puts the loads in
// flight — the atlas dword for the lanes outside, the two brick windows for the lanes inside —, mcm_sample_finish blends them, so
// that both kinds fly together (a wave-event of a HIT tile is a chain of dependent latencies: -1 memory latency per event), and
// the EARLY event loops below put the out-of-cube lanes' path end between the two.  The second phase tests an opaque copy of the
// predicate, or the compiler would thread the phases back into one branch.
// ---- instrumented build (make EXTRA=-DVPT_EVENT_TIMING; tools/r04_event_timing.py): a wave clock read at wave-uniform points of the event
// loop of k_mcm_integrate.  Each mark first waits for the counters it names (so a phase ends when its results are there) and pins the values
// the phase produced (so the compiler cannot move the phase's arithmetic behind the mark); the 100 MHz wall clock (s_memrealtime) is summed
// per phase and wave; every wave adds its sums to ITS OWN 128-byte slot of the timing buffer (PassArgs.violations in such a build:
// [wave of the launch][16]; same-address atomics would serialise at ~12 ns each and stall the memory pipe they measure), the host sums.
// Reading the clock is a scalar memory operation whose round trip lands in the phase that follows the mark: slot 7 holds one extra mark per
// event (two reads back to back), which the tool subtracts from every phase.
// The marks serialise what the shipped kernel overlaps (the wheel draw under the sample's flight, ...): the instrumented kernel is a few
// per cent slower, and its phases are an upper bound of the shipped kernel's.
#ifdef VPT_EVENT_TIMING
#define VPT_TIMING_WAVES 16384
struct EventClock { unsigned long long prev, start; unsigned long long acc[8]; };
VPT_DEV void ev_start(EventClock &c) { for (int i = 0; i < 8; i++) c.acc[i] = 0ull; c.prev = wall_clock64(); c.start = c.prev; }
#define EV_MARK(c, slot, WAIT) do { asm volatile(WAIT ::: "memory"); const unsigned long long n_ = wall_clock64(); (c).acc[slot] += n_ - (c).prev; (c).prev = n_; } while (0)
#define EV_PIN3(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z))
VPT_DEV void ev_flush(const EventClock &c, unsigned long long *out) {
    if (((int)threadIdx.x & 63) == 0 && out) {
        unsigned long long *slot = out + (((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) & (VPT_TIMING_WAVES - 1)) * 16;
        for (int i = 0; i < 8; i++) slot[i] += c.acc[i];
        slot[8] += 1ull;
        slot[9] = c.start; slot[10] = c.prev;                 // this launch's wave: first and last clock read (absolute)
    }
}
#define EV_PARAM , EventClock &evc
#define EV_ARG , evc
#define EV_LOCAL EventClock evc; ev_start(evc);
#else
#define EV_MARK(c, slot, WAIT) do { } while (0)
#define EV_PIN3(v) do { } while (0)
#define EV_PARAM
#define EV_ARG
#define EV_LOCAL
#endif
struct SampleLoads { uint32_t aw; uint64_t w0, w1; float f0, f1, f2; uint32_t atlas; };
template <int V>
VPT_DEV SampleLoads mcm_sample_issue(const PassArgs &a, const LdsTables &t, f3 p, bool oob) {
    constexpr bool WIDE = (V & VPT_V_WIDE) != 0;
    SampleLoads s;
    s.aw = 0u; s.w0 = 0ull; s.w1 = 0ull; s.f0 = 0.0f; s.f1 = 0.0f; s.f2 = 0.0f;
    // The destination registers of BOTH kinds of load get their zeros HERE, before either load is issued (round 4).  Left to itself the compiler
    // sank `w0 = w1 = 0` into the out-of-cube side and re-used those registers there as scratch: in a divergent wave that side then had to
    // wait (s_waitcnt vmcnt) for the in-cube lanes' brick windows before it could even compute its atlas address — the two gathers this
    // function exists to overlap ran one after the other.
    asm volatile("" : "+v"(s.w0), "+v"(s.w1), "+v"(s.aw));
    const bool at = oob && a.vol.atlas != nullptr;
    if (at) {
        s.aw = a.vol.atlas[boundary_cell(a.vol, p, s.f0, s.f1)];
    } else {
        uint32_t x, y, z;
        linear_cell(p.x, a.vol.fnx, a.vol.hx, x, s.f0);
        linear_cell(p.y, a.vol.fny, a.vol.hy, y, s.f1);
        linear_cell(p.z, a.vol.fnz, a.vol.hz, z, s.f2);
        if (V & VPT_V_REC) {
            s.w0 = record_load(record_addr<WIDE>(a.vol, t, x, y, z));      // records z | z + 1 of the column: all eight taps
        } else {
            const uint8_t *b = cell_addr<WIDE>(a.vol, t, x, y, z);
            __builtin_memcpy(&s.w0, b, 8);
            __builtin_memcpy(&s.w1, b + 25, 8);
        }
    }
    s.atlas = at ? 1u : 0u;
    asm volatile("" : "+v"(s.atlas));
    return s;
}
template <int V>
VPT_DEV float4 mcm_sample_finish(const PassArgs &a, const LdsTables &t, const SampleLoads &s) {
    float r;
    // (the brick windows were issued first and return first: their blend runs while the atlas dword is still on its way)
    if (!s.atlas) {
        if (V & VPT_V_REC) r = record_blend((uint32_t)s.w0, (uint32_t)(s.w0 >> 32), s.f0, s.f1, s.f2);
        else r = trilinear_blend((uint32_t)s.w0, (uint32_t)(s.w0 >> 32), (uint32_t)s.w1, (uint32_t)(s.w1 >> 32), s.f0, s.f1, s.f2);
    } else {
        r = boundary_blend(s.aw, s.f0, s.f1);
    }
    float4 vs = sample_tf(t.tf, a.tf_fw, a.tf_hi, r);
    asm volatile("" : "+v"(vs.w));
    return vs;
}
template <int V>
VPT_DEV float4 mcm_sample(const PassArgs &a, const LdsTables &t, f3 p, bool oob) {
    if (!(V & (VPT_V_NEAREST | VPT_V_RG | VPT_V_F32))) return mcm_sample_finish<V>(a, t, mcm_sample_issue<V>(a, t, p, oob));
    float4 vs = sample_volume_color<V>(a, t, p);
    asm volatile("" : "+v"(vs.w));
    return vs;
}
template <int V>
VPT_DEV void mcm_events(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py EV_PARAM) {
    const f3 from0 = unproject_near(px, py, a);

    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    EV_MARK(evc, 5, "s_waitcnt vmcnt(0) lgkmcnt(0)");                                        // prologue: state load, LDS staging, seed
    for (uint32_t s = 0u; s < a.steps; s++) {
        EV_MARK(evc, 7, "");                                                                   // calibration: the cost of a mark itself
        float dist = random_exponential(state, a.inv_extinction);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        // any(greaterThan(pos, 1)) || any(lessThan(pos, 0)), NaN components compare false
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
#ifdef VPT_EVENT_TIMING
        EV_PIN3(q); EV_MARK(evc, 0, "");                                                      // free path: PCG, log, move, bounds test
        const SampleLoads ld_ = mcm_sample_issue<V>(a, t, q, oob);
        EV_MARK(evc, 1, "s_waitcnt lgkmcnt(0)");                                              // filter cell, LDS address tables, loads issued
        EV_MARK(evc, 2, "s_waitcnt vmcnt(0)");                                                // the loads' flight (atlas dword | brick windows)
        float4 vs = mcm_sample_finish<V>(a, t, ld_);
        EV_MARK(evc, 3, "s_waitcnt lgkmcnt(0)");                                              // blend + transfer function (LDS)
#else
        float4 vs = mcm_sample<V>(a, t, q, oob);
#endif
        float p_null = 1.0f - vs.w;
        float p_scat = (ph.bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        float wheel = random_uniform(state);
        if (oob || wheel < p_abs) {
            // out of bounds: radiance = transmittance * env; absorption: radiance = 0 — one shared deposit + resetPhoton
            f3 rad = { 0.0f, 0.0f, 0.0f };
            if (oob) {
                float4 env = sample_environment(a.env, ph.direction);
                rad = f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z };
            }
            photon_deposit(ph, rad);
            reset_photon(state, ph, px, py, a, from0);
        } else if (wheel < p_abs + p_scat) {
            ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
            ph.direction = sample_hg(state, a.anisotropy, ph.direction);
            ph.bounces++;
        }
        EV_PIN3(ph.position); EV_PIN3(ph.direction); EV_MARK(evc, 4, "");                     // wheel, probabilities, path end | scattering
    }
}
// The same events with the out-of-cube lanes' path end (deposit + resetPhoton: they need the random stream only, and whether a
// position is out of bounds is known before its sample) placed BETWEEN issuing the sample's loads and consuming them.  Same draws in
// the same order, same arithmetic: bit-identical.  It costs ~12 more live registers, so it is the form of the HIT-tile kernel where
// occupancy is not what limits it — a shard's few tiles, whose pass is one wave per SIMD walking a chain of dependent latencies
// (DESIGN.md section 8).  LINEAR one-channel byte volumes only (the tile classes' precondition).
template <int V>
VPT_DEV void mcm_events_early(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py) {
    const f3 from0 = unproject_near(px, py, a);
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = random_exponential(state, a.inv_extinction);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        const SampleLoads ld = mcm_sample_issue<V>(a, t, q, oob);
        const uint32_t bounces = ph.bounces;                       // (the scattering probability reads the count before the path end)
        float wheel = random_uniform(state);
        if (oob) {
            float4 env = sample_environment(a.env, ph.direction);
            photon_deposit(ph, f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z });
            reset_photon(state, ph, px, py, a, from0);
        }
        float4 vs = mcm_sample_finish<V>(a, t, ld);
        float p_null = 1.0f - vs.w;
        float p_scat = (bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        if (!oob) {
            if (wheel < p_abs) {
                photon_deposit(ph, f3{ 0.0f, 0.0f, 0.0f });
                reset_photon(state, ph, px, py, a, from0);
            } else if (wheel < p_abs + p_scat) {
                ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
                ph.direction = sample_hg(state, a.anisotropy, ph.direction);
                ph.bounces++;
            }
        }
    }
}
// ---- fast-arithmetic variant of the MCM events (VPT_OPTION_FAST_MATH, kernel variant bit VPT_V_FAST) ---------------------
// The same shader (MCMRenderer.glsl:128-166, resetPhoton :70-78, HG :91-106) with the arithmetic a GPU driver gives GLSL:
// v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_log_f32 / v_sin_f32 / v_cos_f32 (1 ulp class, GLSL ES 3.00 §4.5.1 allows 2.5 ulp for a/b and
// leaves log / sin / cos implementation-defined) instead of the contract's software routines, and algebraically equal forms
// that need fewer instructions:
//   * -log(u)/rate = log2(u) * (-ln 2 / rate);
//   * (random_square * 2 - 1) * inverseResolution = k * (2^-31 / W) - 1 / W with k the PCG state as a float;
//   * inverseMvp * (p + jitter, 1, 1) = inverseMvp * (p, 1, 1) + jitter.x * column0 + jitter.y * column1 (the first term is a
//     pixel constant);
//   * normalize(to.xyz / to.w - from) = sign(to.w) * normalize(to.xyz - to.w * from): no division by w.
// The integer PCG stream is identical, so the two variants take the same decisions except where a comparison falls
// within rounding error; there is NO bit-exact CPU twin of this variant — it is checked against the contract oracle by
// first-event agreement and converged-image statistics (tests/test_gpu_fast_math.py, tolerance in DESIGN.md §3).
VPT_DEV float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }        // 1/(+-0) = +-inf, as the slab test needs
VPT_DEV float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
VPT_DEV float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
VPT_DEV float hw_log2(float x) { return __builtin_amdgcn_logf(x); }       // log2(0) = -inf
VPT_DEV float pcg_float(uint32_t &state) { state = pcg(state); return (float)state; }
// sampleHenyeyGreenstein with hardware sqrt / sin / cos; v_sin_f32 / v_cos_f32 take their argument in revolutions:
// sin(TWOPI * u) = v_sin_f32(u)
VPT_DEV f3 sample_hg_fast(uint32_t &state, float g, f3 dir) {
    float u1 = pcg_float(state) * 0x1p-32f, u2 = pcg_float(state) * 0x1p-32f;
    float r0 = hw_sqrt(u1);
    f2 d = { r0 * __builtin_amdgcn_cosf(u2), r0 * __builtin_amdgcn_sinf(u2) };
    float norm = fmaf(d.y, d.y, d.x * d.x);
    float radius = 2.0f * hw_sqrt(1.0f - norm);
    f3 u = { radius * d.x, radius * d.y, fmaf(-2.0f, norm, 1.0f) };
    if (fabsf(g) < 1e-5f) return u;
    float g2 = g * g;
    float c = (1.0f - g2) * hw_rcp(fmaf(2.0f * g, pcg_float(state) * 0x1p-32f, 1.0f - g));
    float hgcos = fmaf(-c, c, 1.0f + g2) * hw_rcp(2.0f * g);
    float ud = dot3(u, dir);
    f3 cc = { fmaf(-ud, dir.x, u.x), fmaf(-ud, dir.y, u.y), fmaf(-ud, dir.z, u.z) };
    float sq = hw_sqrt(fmaf(-hgcos, hgcos, 1.0f)) * hw_rsq(dot3(cc, cc));
    return f3{ fmaf(sq, cc.x, hgcos * dir.x), fmaf(sq, cc.y, hgcos * dir.y), fmaf(sq, cc.z, hgcos * dir.z) };
}
// the end of a path in the fast variant: deposit `rad`, then resetPhoton (MCMRenderer.glsl:146-151 / 153-158, :70-78)
struct FastPixel { f3 from0; float4 fb; float jx, jy; };
// photon_start in the fast variant's arithmetic: min((0 - f) * iv, (1 - f) * iv) = -f * iv + min(iv, 0), one min and one fma per slab
VPT_DEV f3 photon_start_fast(f3 from0, f3 dir) {
    f3 iv = { hw_rcp(dir.x), hw_rcp(dir.y), hw_rcp(dir.z) };
    float tx = fmaf(-from0.x, iv.x, vmin(iv.x, 0.0f));
    float ty = fmaf(-from0.y, iv.y, vmin(iv.y, 0.0f));
    float tz = fmaf(-from0.z, iv.z, vmin(iv.z, 0.0f));
    float tnear = vmax(vmax(vmax(tx, ty), tz), 0.0f);
    return madd3(from0, tnear, dir);
}
// the pixel constants of the fast variant's resetPhoton with blur == 0: the near-plane point and the far-plane base point (homogeneous)
VPT_DEV FastPixel fast_pixel(const PassArgs &a, float px, float py) {
    FastPixel c;
    const float4 nb = mat4_mul_point(a.mvp_inv, px, py, -1.0f);
    const float inw = hw_rcp(nb.w);
    c.from0 = f3{ nb.x * inw, nb.y * inw, nb.z * inw };
    c.fb = mat4_mul_point(a.mvp_inv, px, py, 1.0f);
    c.jx = 0x1p-31f * a.inv_w; c.jy = 0x1p-31f * a.inv_h;
    return c;
}
template <bool NOBLUR = false>
VPT_DEV void fast_path_end(const PassArgs &a, const FastPixel &c, uint32_t &state, Photon &ph, f3 rad, float px, float py) {
    const float *m = a.mvp_inv.m;
    ph.samples++;
    float inv_n = hw_rcp((float)ph.samples);
    ph.radiance.x = fmaf(rad.x - ph.radiance.x, inv_n, ph.radiance.x);
    ph.radiance.y = fmaf(rad.y - ph.radiance.y, inv_n, ph.radiance.y);
    ph.radiance.z = fmaf(rad.z - ph.radiance.z, inv_n, ph.radiance.z);
    if (NOBLUR || a.blur == 0.0f) {
        state = pcg(pcg(state));                          // the disk sample's two draws (multiplied by blur = 0)
        float ax = fmaf(pcg_float(state), c.jx, -a.inv_w);
        float ay = fmaf(pcg_float(state), c.jy, -a.inv_h);
        float4 th = { fmaf(m[4], ay, fmaf(m[0], ax, c.fb.x)), fmaf(m[5], ay, fmaf(m[1], ax, c.fb.y)),
                      fmaf(m[6], ay, fmaf(m[2], ax, c.fb.z)), fmaf(m[7], ay, fmaf(m[3], ax, c.fb.w)) };
        f3 d = { fmaf(-th.w, c.from0.x, th.x), fmaf(-th.w, c.from0.y, th.y), fmaf(-th.w, c.from0.z, th.z) };
        float inv = __builtin_copysignf(hw_rsq(dot3(d, d)), th.w);
        f3 dir = { d.x * inv, d.y * inv, d.z * inv };
        ph.direction = dir;
        ph.position = photon_start_fast(c.from0, dir);
        ph.bounces = 0u;
        ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
    } else {
        reset_photon(state, ph, px, py, a, c.from0);      // depth-of-field runs: the contract's general path
    }
}
template <int V>
VPT_DEV void mcm_events_fast(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py EV_PARAM) {
    const FastPixel c = fast_pixel(a, px, py);
    // -ln(u * 2^-32) / extinction = (log2(u) - 32) * ld
    const float ld = -0.6931471805599453f * a.inv_extinction, ld32 = -32.0f * ld;

    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    EV_MARK(evc, 5, "s_waitcnt vmcnt(0) lgkmcnt(0)");
    for (uint32_t s = 0u; s < a.steps; s++) {
        EV_MARK(evc, 7, "");
        float dist = fmaf(hw_log2(pcg_float(state)), ld, ld32);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
#ifdef VPT_EVENT_TIMING
        EV_PIN3(q); EV_MARK(evc, 0, "");
        const SampleLoads ld_ = mcm_sample_issue<V>(a, t, q, oob);
        EV_MARK(evc, 1, "s_waitcnt lgkmcnt(0)");
        EV_MARK(evc, 2, "s_waitcnt vmcnt(0)");
        float4 vs = mcm_sample_finish<V>(a, t, ld_);
        EV_MARK(evc, 3, "s_waitcnt lgkmcnt(0)");
#else
        float4 vs = mcm_sample<V>(a, t, q, oob);
#endif
        float p_null = 1.0f - vs.w;
        float p_scat = (ph.bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        float wheel = pcg_float(state) * 0x1p-32f;
        if (oob || wheel < p_abs) {
            f3 rad = { 0.0f, 0.0f, 0.0f };
            if (oob) {
                float4 env = sample_environment(a.env, ph.direction);
                rad = f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z };
            }
            fast_path_end(a, c, state, ph, rad, px, py);
        } else if (wheel < p_abs + p_scat) {
            ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
            ph.direction = sample_hg_fast(state, a.anisotropy, ph.direction);
            ph.bounces++;
        }
        EV_PIN3(ph.position); EV_PIN3(ph.direction); EV_MARK(evc, 4, "");
    }
}
template <int V>
VPT_DEV void mcm_events_fast_early(const PassArgs &a, const LdsTables &t, Photon &ph, float px, float py) {     // see mcm_events_early
    const FastPixel c = fast_pixel(a, px, py);
    const float ld = -0.6931471805599453f * a.inv_extinction, ld32 = -32.0f * ld;
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = fmaf(hw_log2(pcg_float(state)), ld, ld32);
        ph.position = madd3(ph.position, dist, ph.direction);
        f3 q = ph.position;
        bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        const SampleLoads lds = mcm_sample_issue<V>(a, t, q, oob);
        const uint32_t bounces = ph.bounces;
        float wheel = pcg_float(state) * 0x1p-32f;
        if (oob) {
            float4 env = sample_environment(a.env, ph.direction);
            fast_path_end(a, c, state, ph, f3{ ph.transmittance.x * env.x, ph.transmittance.y * env.y, ph.transmittance.z * env.z }, px, py);
        }
        float4 vs = mcm_sample_finish<V>(a, t, lds);
        float p_null = 1.0f - vs.w;
        float p_scat = (bounces >= a.max_bounces) ? 0.0f : vs.w * vmax(vmax(vs.x, vs.y), vs.z);
        float p_abs = 1.0f - p_null - p_scat;
        if (!oob) {
            if (wheel < p_abs) {
                fast_path_end(a, c, state, ph, f3{ 0.0f, 0.0f, 0.0f }, px, py);
            } else if (wheel < p_abs + p_scat) {
                ph.transmittance.x *= vs.x; ph.transmittance.y *= vs.y; ph.transmittance.z *= vs.z;
                ph.direction = sample_hg_fast(state, a.anisotropy, ph.direction);
                ph.bounces++;
            }
        }
    }
}
// Photon state in HBM (MCMRenderer.js:214-263 keeps four RGBA32F attachments: [pos, 0] [dir, bounces] [T, 0] [radiance, samples]):
// the two constant zeros are not stored — position and transmittance are 12-byte texels (one dwordx3 per lane, a wave's
// 64 texels one contiguous 768-byte segment), direction+bounces and radiance+samples 16-byte texels: 56 bytes per pixel each
// way instead of 64.  vpt_renderer_read re-expands them to RGBA32F (k_detile_mcm3).
struct PhotonState { f3 s0; float4 s1; f3 s2; float4 s3; };
VPT_DEV PhotonState photon_load(const PassArgs &a, int k) {
    PhotonState s;
    s.s0 = ((const f3 *)a.st0)[k]; s.s1 = a.st1[k]; s.s2 = ((const f3 *)a.st2)[k]; s.s3 = a.st3[k];
    return s;
}
VPT_DEV void photon_store(const PassArgs &a, int k, const struct Photon &ph);
VPT_DEV Photon photon_unpack(float4 s0, float4 s1, float4 s2, float4 s3) {   // MCMRenderer.glsl:117-126
    Photon ph;
    ph.position = f3{ s0.x, s0.y, s0.z };
    ph.direction = f3{ s1.x, s1.y, s1.z };
    ph.bounces = (uint32_t)(s1.w + 0.5f);
    ph.transmittance = f3{ s2.x, s2.y, s2.z };
    ph.radiance = f3{ s3.x, s3.y, s3.z };
    ph.samples = (uint32_t)(s3.w + 0.5f);
    return ph;
}
VPT_DEV Photon photon_unpack(const PhotonState &s) {
    return photon_unpack(make_float4(s.s0.x, s.s0.y, s.s0.z, 0.0f), s.s1, make_float4(s.s2.x, s.s2.y, s.s2.z, 0.0f), s.s3);
}
VPT_DEV void photon_store(const PassArgs &a, int k, const Photon &ph) {       // MCMRenderer.glsl:168-171
    ((f3 *)a.st0)[k] = ph.position;
    a.st1[k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, (float)ph.bounces);
    ((f3 *)a.st2)[k] = ph.transmittance;
    a.st3[k] = make_float4(ph.radiance.x, ph.radiance.y, ph.radiance.z, (float)ph.samples);
}

// Persistent form of the integrate pass: every wave walks several 8x8-pixel segments of the tile-ordered state arrays
// (segment g = lanes [64g, 64g+64)) and loads the NEXT segment's photon state (4 x dwordx4 per lane) before it starts the
// current segment's events; LDS tables are staged once per workgroup instead of once per tile.  Measured (512^3, 1080p,
// steps 8): 0.156 ms vs 0.148 ms for the one-workgroup-per-tile kernel — the prefetch registers cost two waves per SIMD
// (96 vs 72 VGPRs) and the state stream is only ~20 us of the frame (timing builds without state loads / stores:
// -4 us / -13 us), so this form is kept as an option (VPT_OPTION_MCM_PERSISTENT), not the default.
template <bool FUSE_RENDER, int V, bool PREFETCH>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(PREFETCH ? 5 : 7, 8))) k_mcm_persist(PassArgs a, int nseg) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0, (V & VPT_V_REC) != 0>(lds_raw, a);
    const int lane = (int)threadIdx.x & 63;
    const int nwaves = (int)gridDim.x * (VPT_BLOCK / 64);
    int g = (int)blockIdx.x * (VPT_BLOCK / 64) + ((int)threadIdx.x >> 6);
    if (g >= nseg) return;
    size_t k = (size_t)g * 64 + lane;
    PhotonState nx;
    if (PREFETCH) nx = photon_load(a, (int)k);
    while (g < nseg) {
        const int gc = g;
        const size_t kc = k;
        if (!PREFETCH) nx = photon_load(a, (int)k);
        Photon ph = photon_unpack(nx);
        g += nwaves;
        k = (size_t)g * 64 + lane;
        if (PREFETCH && g < nseg) {              // wave-uniform: prefetch the next segment's photon state
            nx = photon_load(a, (int)k);
        }
        int t16 = gc >> 2, w = gc & 3;
        int ty = t16 / a.pm.tiles_x, tx = t16 - ty * a.pm.tiles_x;
        int i = tx * VPT_TILE + (w & 1) * 8 + (lane & 7);
        int l = ty * VPT_TILE + (w >> 1) * 8 + (lane >> 3);
        int j = global_row(a.pm, l);
        if (i < a.pm.W && l < a.pm.local_h && j < a.pm.H) {
            EV_LOCAL
            mcm_events<V>(a, t, ph, ndc_col(a.pm, i), ndc_row(a.pm, j) EV_ARG);
            photon_store(a, (int)kc, ph);
            if (FUSE_RENDER)
                a.render[(size_t)l * a.pm.W + i] = pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f);
        }
    }
}


#ifndef VPT_MCM_RG_WAVES
#define VPT_MCM_RG_WAVES 6
#endif
#ifndef VPT_MULTI_WAVES
#define VPT_MULTI_WAVES 7
#endif
#ifndef VPT_MCM_WAVES
#define VPT_MCM_WAVES 7          // waves per SIMD the integrate kernel is compiled for (72 VGPRs; 8 needs 64: A/B in DESIGN.md section 5)
#endif
// integrate/fragment main(): MCMRenderer.glsl:116-172.  FUSE_RENDER additionally performs
// _renderFrame (MCMRenderer.glsl:204-206) on the radiance it just produced.
template <bool FUSE_RENDER, int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu((V & VPT_V_RG) ? VPT_MCM_RG_WAVES : VPT_MCM_WAVES, 8))) k_mcm_integrate(PassArgs a) {
#ifdef VPT_EVENT_TIMING
    EventClock evc; ev_start(evc);
#endif
    apply_frame_table(a);
    // the photon state (4 x dwordx4 per lane, one contiguous 1 KiB segment per wave and array) does not depend on the LDS
    // image: its loads are issued first, so they fly while the workgroup stages the tables and hashes its seed
    Pix p = map_pixel(a.pm);
    PhotonState st;
    if (p.tile) st = photon_load(a, p.k);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0, (V & VPT_V_REC) != 0>(lds_raw, a);
    if (!p.valid) return;
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(st);
    if (V & VPT_V_FAST) mcm_events_fast<V & ~VPT_V_FAST>(a, t, ph, px, py EV_ARG);
    else mcm_events<V>(a, t, ph, px, py EV_ARG);
    photon_store(a, p.k, ph);
    if (FUSE_RENDER) store_frame(a, p, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
#ifdef VPT_EVENT_TIMING
    EV_MARK(evc, 6, "s_waitcnt vmcnt(0)");                                                    // epilogue: state and frame stores
    ev_flush(evc, a.violations);
#endif
}
// the HIT-tile kernel with the early path end (mcm_events_early), compiled for 5 waves per SIMD: selected by the library for tile lists
// short enough to be resident at once at that occupancy (shards)
template <bool FUSE_RENDER, int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8))) k_mcm_integrate_early(PassArgs a) {
    apply_frame_table(a);
    Pix p = map_pixel(a.pm);
    PhotonState st;
    if (p.tile) st = photon_load(a, p.k);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0, (V & VPT_V_REC) != 0>(lds_raw, a);
    if (!p.valid) return;
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(st);
    if (V & VPT_V_FAST) mcm_events_fast_early<V & ~VPT_V_FAST>(a, t, ph, px, py);
    else mcm_events_early<V>(a, t, ph, px, py);
    photon_store(a, p.k, ph);
    if (FUSE_RENDER) store_frame(a, p, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
// =============================================================================================
// Tile classes (round 3).  The host classifies every 16x16 tile against the cube once per reset (vpt_hip.hip classify_tiles:
// a conservative test of the tile's pixel columns, jitter included, against the projected cube, enlarged): a MISS tile is one
// none of whose camera rays can meet the cube.  resetPhoton (MCMRenderer.glsl:70-78) parks such a photon at from + tnear * dir,
// outside the cube, on a ray that leaves it behind — so EVERY event of a MISS tile's pixel is "sample at the clamped position
// (MCMRenderer.glsl:132), find the position out of bounds (:135), deposit transmittance * environment (:136-140), resetPhoton
// (:141)": no in-cube sample, no absorption / scattering / null branch, and at the end of every pass transmittance = (1, 1, 1),
// bounces = 0 and position = photon_start(from0, direction).  k_mcm_miss runs exactly that straight-line event — the same
// arithmetic, draw for draw, as the oob lanes of k_mcm_integrate, so the buffers are bit-identical — with
//   * the sample still executed for every event (boundary atlas gather + transfer-function lookup, result kept alive),
//   * 32 instead of 56 bytes of photon state each way: only [direction, bounces] and [radiance, samples] are read and written; the
//     position is recomputed from the direction, and the position / transmittance arrays of MISS tiles are brought up to date by
//     k_mcm_materialize before anything else looks at them (read-back, whole-image kernels, a changed matrix),
//   * no brick tables in LDS (the atlas needs none), fewer registers: 8 waves per SIMD.
// At the benchmark camera ~78 % of the tiles are MISS tiles.  HIT tiles run k_mcm_integrate from their own tile list.
// CHECK: count the events that contradict the classification (tests / VPT_OPTION_VERIFY_TILE_CLASSES; must stay 0).
// =============================================================================================
VPT_DEV const float4 *stage_tf(float4 *lds, const PassArgs &a) {
    const int nthreads = (int)blockDim.x;
    for (int t = (int)threadIdx.x; t < a.tf_w; t += nthreads) {
        float4 v = a.tf[t], n = a.tf[min(t + 1, a.tf_w - 1)];
        lds[2 * t] = v;
        lds[2 * t + 1] = make_float4(n.x - v.x, n.y - v.y, n.z - v.z, n.w - v.w);
    }
    __syncthreads();
    return lds;
}
// the executed-and-discarded sample of an out-of-cube event: texture(uVolume, clamp(p)) through the boundary atlas, then the
// transfer function (MCMRenderer.glsl:85-89,132).  Precondition (the tile class): p has a coordinate outside [0, 1].
// In two phases, like mcm_sample: miss_sample_issue puts the atlas gather in flight, miss_sample_finish blends it and looks the
// transfer function up; the event's path end (which needs the random stream only) sits between the two, under the load's latency.
struct MissLoad { uint32_t aw; float fa, fb; };
template <bool CHECK>
VPT_DEV MissLoad miss_sample_issue(const PassArgs &a, f3 q, unsigned long long *violations) {
    MissLoad l;
    l.aw = a.vol.atlas[boundary_cell(a.vol, q, l.fa, l.fb)];
    if (CHECK) {
        const bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        if (!oob) atomicAdd(violations, 1ull);
    }
    return l;
}
VPT_DEV void miss_sample_finish(const PassArgs &a, const float4 *tf, const MissLoad &l) {
    float4 vs = sample_tf(tf, a.tf_fw, a.tf_hi, boundary_blend(l.aw, l.fa, l.fb));
    asm volatile("" : "+v"(vs.w));
}
// The same executed-and-discarded sample for the other volume formats (round 4): NEAREST filter, two channels (RG8 / RG32F: the G atlas 6 faces
// behind the R atlas, the transfer function looked up in 2-D from HBM), float texels (one float4 per cell, no normalisation).  Same texels,
// same order of operations as sample_volume_rg<V> at the clamped position: bit-identical (float texels: finite ones — the library does not
// use the atlas of a float volume that holds others, vpt_volume_finalize).  One phase: these variants are not the benchmark's.
template <int V>
VPT_DEV void miss_sample_any(const PassArgs &a, const float4 *tf, f3 q) {
    const f2 rg = sample_boundary_rg<V>(a.vol, q);
    float4 vs = (V & VPT_V_RG) ? sample_tf2d(a.tf, a.tf_w, a.tf_h, rg.x, rg.y) : sample_tf(tf, a.tf_fw, a.tf_hi, rg.x);
    asm volatile("" : "+v"(vs.w));
    if (a.miss_verify) {
        const bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
        if (!oob) atomicAdd(a.violations, 1ull);
    }
}
// LATE: the sample is consumed after the path end (under whose arithmetic its load flies) instead of right where the shader samples
template <int V, bool CHECK, bool LATE>
VPT_DEV void mcm_events_miss(const PassArgs &a, const float4 *tf, Photon &ph, float px, float py, f3 from0) {
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = random_exponential(state, a.inv_extinction);
        ph.position = madd3(ph.position, dist, ph.direction);
        constexpr bool OTHER = (V & (VPT_V_NEAREST | VPT_V_RG | VPT_V_F32)) != 0;      // another volume format: the one-phase sample
        MissLoad l = { 0u, 0.0f, 0.0f };
        if (OTHER) miss_sample_any<V>(a, tf, ph.position);
        else l = miss_sample_issue<CHECK>(a, ph.position, a.violations);
        if (!LATE && !OTHER) miss_sample_finish(a, tf, l);
        random_uniform(state);                                     // the wheel draw (its value decides nothing out of bounds)
        float4 env = sample_environment(a.env, ph.direction);      // transmittance is (1, 1, 1): radiance = 1 * env, exactly env
        photon_deposit(ph, f3{ env.x, env.y, env.z });
        reset_photon<true>(state, ph, px, py, a, from0);
        if (LATE && !OTHER) miss_sample_finish(a, tf, l);
    }
}
template <int V, bool CHECK, bool LATE>
VPT_DEV void mcm_events_miss_fast(const PassArgs &a, const float4 *tf, const FastPixel &c, Photon &ph, float px, float py) {
    const float ld = -0.6931471805599453f * a.inv_extinction, ld32 = -32.0f * ld;
    uint32_t state = hash3(__float_as_uint(ndc_to_uv(px)), __float_as_uint(ndc_to_uv(py)), __float_as_uint(a.seed));
    for (uint32_t s = 0u; s < a.steps; s++) {
        float dist = fmaf(hw_log2(pcg_float(state)), ld, ld32);
        ph.position = madd3(ph.position, dist, ph.direction);
        constexpr bool OTHER = (V & (VPT_V_NEAREST | VPT_V_RG | VPT_V_F32)) != 0;
        MissLoad l = { 0u, 0.0f, 0.0f };
        if (OTHER) miss_sample_any<V>(a, tf, ph.position);
        else l = miss_sample_issue<CHECK>(a, ph.position, a.violations);
        if (!LATE && !OTHER) miss_sample_finish(a, tf, l);
        state = pcg(state);                                        // the wheel draw
        float4 env = sample_environment(a.env, ph.direction);
        fast_path_end<true>(a, c, state, ph, f3{ env.x, env.y, env.z }, px, py);
        if (LATE && !OTHER) miss_sample_finish(a, tf, l);
    }
}
// (8 waves per SIMD = 64 VGPRs; the contract arithmetic of the other volume formats needs two more: 7 waves there instead of a spill)
template <bool FUSE_RENDER, int V, bool CHECK, bool LATE>
__global__ void __launch_bounds__(VPT_BLOCK)
__attribute__((amdgpu_waves_per_eu(((V & (VPT_V_NEAREST | VPT_V_RG | VPT_V_F32)) && !(V & VPT_V_FAST)) ? 7 : 8, 8))) k_mcm_miss(PassArgs a) {
    apply_frame_table(a);
    Pix p = map_pixel(a.pm);
    float4 s1 = make_float4(0.0f, 0.0f, 1.0f, 0.0f), s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f3 s0 = { 0.0f, 0.0f, 0.0f };
    if (p.tile) {
        s1 = a.st1[p.k]; s3 = a.st3[p.k];
        if (a.miss_load_pos) s0 = ((const f3 *)a.st0)[p.k];
    }
    extern __shared__ float4 lds_raw[];
    const float4 *tf = stage_tf(lds_raw, a);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph;
    ph.direction = f3{ s1.x, s1.y, s1.z };
    ph.bounces = 0u;
    ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
    ph.radiance = f3{ s3.x, s3.y, s3.z };
    ph.samples = (uint32_t)(s3.w + 0.5f);
    if (V & VPT_V_FAST) {
        const FastPixel c = fast_pixel(a, px, py);
        ph.position = a.miss_load_pos ? s0 : photon_start_fast(c.from0, ph.direction);
        mcm_events_miss_fast<V & ~VPT_V_FAST, CHECK, LATE>(a, tf, c, ph, px, py);
    } else {
        const f3 from0 = unproject_near(px, py, a);
        ph.position = a.miss_load_pos ? s0 : photon_start(from0, ph.direction);
        mcm_events_miss<V, CHECK, LATE>(a, tf, ph, px, py, from0);
    }
    a.st1[p.k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, 0.0f);
    a.st3[p.k] = make_float4(ph.radiance.x, ph.radiance.y, ph.radiance.z, (float)ph.samples);
    if (FUSE_RENDER) store_frame(a, p, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
// brings the position / transmittance arrays of the MISS tiles up to date: position = photon_start(from0, direction) in the
// arithmetic of the variant that ran the last pass, transmittance = (1, 1, 1)
template <bool FAST>
__global__ void __launch_bounds__(VPT_BLOCK) k_mcm_materialize(PassArgs a) {
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    const float4 s1 = a.st1[p.k];
    const f3 dir = { s1.x, s1.y, s1.z };
    f3 pos;
    if (FAST) pos = photon_start_fast(fast_pixel(a, px, py).from0, dir);
    else pos = photon_start(unproject_near(px, py, a), dir);
    ((f3 *)a.st0)[p.k] = pos;
    ((f3 *)a.st2)[p.k] = f3{ 1.0f, 1.0f, 1.0f };
}

// `npasses` whole render() passes of one pixel in ONE launch (vpt_renderer_play, VPT_PLAY_FUSED): the photon state
// stays in registers between passes — one 64 B read + 64 B write per pixel for the whole sequence instead of per pass —
// and the launch / staging cost is paid once.  Pass f re-seeds from the f-th entry of the frame table exactly as
// launch f of the unfused sequence would; the render buffer receives the last pass's radiance, which is all that is
// left of the unfused sequence's render buffer as well.
template <int V, bool FRAMES>
VPT_DEV void mcm_multi_body(PassArgs &a, uint32_t npasses, uint2 *ring, uint32_t slot_pixels) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0, (V & VPT_V_REC) != 0>(lds_raw, a);
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(photon_load(a, p.k));
    uint32_t base = a.frame_base;
    for (uint32_t f = 0; f < npasses; f++) {
        a.seed = a.frame_table[(base + f) & a.frame_mask].seed;
        EV_LOCAL
        if (V & VPT_V_FAST) mcm_events_fast<V & ~VPT_V_FAST>(a, t, ph, px, py EV_ARG);
        else mcm_events<V>(a, t, ph, px, py EV_ARG);
        // the unfused sequence stores the counters as floats between passes and re-reads them with uint(w + 0.5):
        // identical for every count below 2^24
        // VPT_PLAY_FRAMES: every pass's frame is written (slot f of the frame ring), as `npasses` render() calls would show them
        if (FRAMES) store_frame_texel(&ring[(size_t)f * slot_pixels + (size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
    }
    photon_store(a, p.k, ph);
    store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(VPT_MULTI_WAVES, 8))) k_mcm_multi(PassArgs a, uint32_t npasses) {
    mcm_multi_body<V, false>(a, npasses, nullptr, 0u);
}
// the same with every pass's frame written to the ring: one more live address per lane (and, since the sample's loads are issued in
// a phase of their own, two more windows in flight): compiled for 4 waves per SIMD (128 VGPRs; at 96 it spills three registers)
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) k_mcm_frames(PassArgs a, uint32_t npasses, uint2 *ring, uint32_t slot_pixels) {
    mcm_multi_body<V, true>(a, npasses, ring, slot_pixels);
}

// ---- a bucket of frames by ONE launch per tile class (VPT_OPTION_BUCKET_KERNEL, vpt_renderer_play_into) --------------------------
// `nframes` render() passes of the tiles of one class, each pass's frame written to its slot of the caller's bucket (slot f =
// ring + f * slot_pixels): what nframes launches of k_mcm_integrate<true> / k_mcm_miss<true> with those render targets write, with
// the photon state in registers from the first pass to the last.  A rank's share of a sharded frame is a few hundred tiles: its pass
// is ~8 us of arithmetic behind ~4-5 us of launch gap, table staging and state traffic (profiles/r03_shard8.json), and a bucket of
// frames that one collective moves anyway (vpt_amd/tiles.py FrameGather) pays those once.  The frames' seeds travel BY VALUE: the two
// classes run on two streams that nothing orders against an upload of the frame table.
struct FrameSeeds { float seed[VPT_BUCKET_FRAMES]; };
// DISPLAY: the slots hold the frames as the armed tone mapper shows them (RGBA8 through its table, PassArgs.tm_table) instead of RGBA16F:
// half the bytes for the collective that moves the bucket (vpt_renderer_play_into_display)
template <bool DISPLAY>
VPT_DEV void bucket_store(const PassArgs &a, void *ring, size_t texel, uint2 v) {
    if (DISPLAY) ((uint32_t *)ring)[texel] = tone_map_texel(a.tm_table, v, nullptr);
    else store_frame_texel((uint2 *)ring + texel, v);
}
template <int V, bool EARLY, bool DISPLAY>
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_mcm_bucket_hit(PassArgs a, FrameSeeds fs, uint32_t nframes, void *ring, uint32_t slot_pixels) {
    Pix p = map_pixel(a.pm);
    PhotonState st;
    if (p.tile) st = photon_load(a, p.k);
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0, (V & VPT_V_REC) != 0>(lds_raw, a);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph = photon_unpack(st);
    size_t texel = (size_t)p.l * a.pm.W + p.i;
    for (uint32_t f = 0; f < nframes; f++) {
        a.seed = fs.seed[f];
        if (V & VPT_V_FAST) {
            if (EARLY) mcm_events_fast_early<V & ~VPT_V_FAST>(a, t, ph, px, py);
            else { EV_LOCAL mcm_events_fast<V & ~VPT_V_FAST>(a, t, ph, px, py EV_ARG); }
        } else {
            if (EARLY) mcm_events_early<V>(a, t, ph, px, py);
            else { EV_LOCAL mcm_events<V>(a, t, ph, px, py EV_ARG); }
        }
        // (between two launches the counters travel as floats and come back through uint(w + 0.5): the identity below 2^24)
        bucket_store<DISPLAY>(a, ring, texel, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
        texel += slot_pixels;
    }
    photon_store(a, p.k, ph);
    // (VPT_PLAY_FRAMES: the render buffer shows the last frame, as after `nframes` render() calls; null for a caller's bucket)
    if (a.render) store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
template <int V, bool LATE, bool DISPLAY>     // (the contract's arithmetic keeps six more values alive across the frame loop: 6 waves per SIMD there)
__global__ void __launch_bounds__(VPT_BLOCK) __attribute__((amdgpu_waves_per_eu((V & VPT_V_FAST) ? 8 : 6, 8)))
k_mcm_bucket_miss(PassArgs a, FrameSeeds fs, uint32_t nframes, void *ring, uint32_t slot_pixels) {
    Pix p = map_pixel(a.pm);
    float4 s1 = make_float4(0.0f, 0.0f, 1.0f, 0.0f), s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    f3 s0 = { 0.0f, 0.0f, 0.0f };
    if (p.tile) {
        s1 = a.st1[p.k]; s3 = a.st3[p.k];
        if (a.miss_load_pos) s0 = ((const f3 *)a.st0)[p.k];
    }
    extern __shared__ float4 lds_raw[];
    const float4 *tf = stage_tf(lds_raw, a);
    if (!p.valid) return;
    const float px = ndc_col(a.pm, p.i), py = ndc_row(a.pm, p.j);
    Photon ph;
    ph.direction = f3{ s1.x, s1.y, s1.z };
    ph.bounces = 0u;
    ph.transmittance = f3{ 1.0f, 1.0f, 1.0f };
    ph.radiance = f3{ s3.x, s3.y, s3.z };
    ph.samples = (uint32_t)(s3.w + 0.5f);
    size_t texel = (size_t)p.l * a.pm.W + p.i;
    if (V & VPT_V_FAST) {
        const FastPixel c = fast_pixel(a, px, py);
        ph.position = a.miss_load_pos ? s0 : photon_start_fast(c.from0, ph.direction);
        for (uint32_t f = 0; f < nframes; f++) {
            a.seed = fs.seed[f];
            mcm_events_miss_fast<V & ~VPT_V_FAST, false, LATE>(a, tf, c, ph, px, py);
            bucket_store<DISPLAY>(a, ring, texel, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
            texel += slot_pixels;
        }
    } else {
        const f3 from0 = unproject_near(px, py, a);
        ph.position = a.miss_load_pos ? s0 : photon_start(from0, ph.direction);
        for (uint32_t f = 0; f < nframes; f++) {
            a.seed = fs.seed[f];
            mcm_events_miss<V, false, LATE>(a, tf, ph, px, py, from0);
            bucket_store<DISPLAY>(a, ring, texel, pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
            texel += slot_pixels;
        }
    }
    a.st1[p.k] = make_float4(ph.direction.x, ph.direction.y, ph.direction.z, 0.0f);
    a.st3[p.k] = make_float4(ph.radiance.x, ph.radiance.y, ph.radiance.z, (float)ph.samples);
    if (a.render) store_frame_texel(&a.render[(size_t)p.l * a.pm.W + p.i], pack_half4(ph.radiance.x, ph.radiance.y, ph.radiance.z, 1.0f));
}
#ifdef VPT_MCM_PLAIN_KERNELS
__global__ void __launch_bounds__(VPT_BLOCK) k_mcm_render(PassArgs a) {   // MCMRenderer.glsl:204-206
    Pix p = map_pixel(a.pm);
    if (!p.valid) return;
    float4 r = a.st3[p.k];
    a.render[(size_t)p.l * a.pm.W + p.i] = pack_half4(r.x, r.y, r.z, 1.0f);
}
#endif
