"""The GLSL interpreter itself (oracle/glsl_interp.py) on small programs written for this test, with answers that follow from the GLSL ES
3.00 specification alone: integer wrap-around and conversions, operator precedence, short-circuit evaluation, swizzles on both sides of an
assignment, struct and array value semantics, in / out / inout copy-in copy-out, overload resolution, loops with break / continue,
matrix-vector products, the built-ins' definitions (mix, mod, clamp, step, smoothstep, min / max with the spec's operand order), texture
filtering at texel centres and half-way between them, CLAMP_TO_EDGE and REPEAT, NEAREST.  What the reference's shaders are held to
(tests/test_glsl_reference.py) is only as good as this."""
import numpy as np
import pytest

from oracle import glsl_interp as G

F = np.float32
HEAD = "#version 300 es\nprecision mediump float;\n"


def run(body, decls="", uniforms=None, outs="out vec4 o;", contract=False):
    src = HEAD + decls + "\n" + outs + "\nvoid main() {\n" + body + "\n}\n"
    sh = G.Shader(src, G.Math(contract=contract))
    return sh.run(uniforms or {}, {})


def floats(v):
    return [float(x) for x in v.c]


def test_integer_arithmetic_and_conversions():
    r = run("""
        uint a = 4000000000u; uint b = a * 3u + 7u;            // wraps modulo 2^32
        uint h = 747796405u * 2891336453u;
        int i = -7; int q = i / 2; int m = i - (i / 2) * 2;    // truncation toward zero
        o = vec4(float(b), float(~0u), float(q), float(m));
        p = uvec4(b, h, uint(3.99), uint(i + 8));
        s = ivec4(int(-2.7), (5 >> 1) | (1 << 4), 7 & 12, 6 ^ 3);
    """, outs="out vec4 o; out uvec4 p; out ivec4 s;")
    assert floats(r["o"]) == [float(F((4000000000 * 3 + 7) % 2**32)), 4294967296.0, -3.0, -1.0]
    assert [x.v for x in r["p"].c] == [(4000000000 * 3 + 7) % 2**32, (747796405 * 2891336453) % 2**32, 3, 1]
    assert [x.v for x in r["s"].c] == [-2, 18, 4, 5]


def test_the_pcg_hash_of_the_contract():
    """pcg.glsl's three lines, written here from the published algorithm, against tests/golden/pcg_kat.json's known answers"""
    r = run("""
        p = uvec4(hash(0u), hash(1u), hash(4294967295u), hash(12345u));
    """, decls="""
        uint hash(uint x) { x = x * 747796405u + 2891336453u; x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u; return (x >> 22u) ^ x; }
    """, outs="out uvec4 p;")
    assert [x.v for x in r["p"].c] == [129708002, 2831084092, 3861530882, 4099845390]


def test_precedence_short_circuit_and_ternary():
    r = run("""
        int n = 0;
        bool a = (n > 0) && (bump(n) > 0);      // right side not evaluated
        bool b = (n == 0) || (bump(n) > 0);     // right side not evaluated
        bool c = (n == 0) && (bump(n) > 0);     // evaluated once
        float t = 2.0 + 3.0 * 4.0 - 6.0 / 3.0;
        float u = -2.0 * -3.0;
        float w = n == 1 ? 10.0 : 20.0;
        o = vec4(t, u, w, float(n) + (a ? 100.0 : 0.0) + (b ? 10.0 : 0.0) + (c ? 1000.0 : 0.0));
    """, decls="int bump(inout int k) { k++; return k; }")
    assert floats(r["o"]) == [12.0, 6.0, 10.0, 1.0 + 10.0 + 1000.0]


def test_swizzles_structs_arrays_and_parameter_qualifiers():
    r = run("""
        vec4 v = vec4(1, 2, 3, 4);
        v.zx = v.xy * 10.0;                      // v = (20, 2, 10, 4)
        v.a += 1.0;
        vec3 w = v.bgr;                          // (10, 2, 20)
        P p; p.pos = w; p.n = 3u;
        P q = p; q.pos.y = 99.0;                 // value semantics: p unchanged
        float arr[3]; arr[0] = 1.0; arr[1] = 2.0; arr[2] = 4.0;
        float outv; float io = 5.0;
        float ret = f(arr[1], outv, io, q);      // in: copy; out: written back; inout: both ways
        o = vec4(p.pos.y, q.pos.y, ret, outv + io);
        o2 = vec4(v);
    """, decls="""
        struct P { vec3 pos; uint n; };
        float f(in float a, out float b, inout float c, P s) { a = a * 2.0; b = a + 1.0; c = c + s.pos.y; s.pos = vec3(0); return a; }
    """, outs="out vec4 o; out vec4 o2;")
    assert floats(r["o"]) == [2.0, 99.0, 4.0, 5.0 + (5.0 + 99.0)]
    assert floats(r["o2"]) == [20.0, 2.0, 10.0, 5.0]


def test_overloads_loops_and_early_exit():
    r = run("""
        float s = 0.0;
        for (uint i = 0u; i < 10u; i++) { if (i == 2u) continue; if (i == 6u) break; s += float(i); }     // 0+1+3+4+5
        float t = 0.0; int k = 0;
        do { t += 0.25; k++; } while (t < 1.0);                                                            // four trips
        float w = 0.0;
        while (true) { w += 1.0; if (w > 2.5) break; }
        o = vec4(s, float(k), w, g(2.0) + g(vec2(1.0, 2.0)) + g(3u));
    """, decls="""
        float g(float x) { return x; }
        float g(vec2 x) { return 10.0 * (x.x + x.y); }
        float g(uint x) { return 100.0 * float(x); }
    """)
    assert floats(r["o"]) == [13.0, 4.0, 3.0, 2.0 + 30.0 + 300.0]


def test_matrices_are_column_major():
    m = G.mat4([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
    r = run("o = m * vec4(1, 0, 2, 1); o2 = vec4(m[1].z, m[3][0], (m * vec4(0, 0, 0, 1)).w, mat2(1.0, 2.0, 3.0, 4.0)[1].x + (mat2(1.0, 2.0, 3.0, 4.0) * vec2(1, 1)).y);",
            decls="uniform mat4 m;", uniforms={"m": m}, outs="out vec4 o; out vec4 o2;")
    assert floats(r["o"]) == [1 + 18 + 13, 2 + 20 + 14, 3 + 22 + 15, 4 + 24 + 16]
    assert floats(r["o2"]) == [7.0, 13.0, 16.0, 3.0 + (2.0 + 4.0)]


def test_builtins_follow_their_definitions():
    r = run("""
        o = vec4(mix(2.0, 10.0, 0.25), mod(-1.0, 3.0), clamp(5.0, 0.0, 1.0), step(0.5, 0.5));
        o2 = vec4(smoothstep(0.0, 2.0, 1.0), fract(-0.25), length(vec3(3, 4, 12)), dot(vec2(1, 2), vec2(3, 4)));
        o3 = vec4(max(vec2(1.0, -1.0), 0.0), min(2.0, 1.0), distance(vec2(0), vec2(3, 4)));
        b = uvec4(floatBitsToUint(1.0), floatBitsToUint(-0.0), any(greaterThan(vec3(0, 2, 0), vec3(1))) ? 1u : 0u, any(lessThan(vec3(0), vec3(0))) ? 1u : 0u);
        vec3 n = normalize(vec3(0, 3, 4));
        o4 = vec4(n, sqrt(2.25));
    """, outs="out vec4 o; out vec4 o2; out vec4 o3; out uvec4 b; out vec4 o4;")
    assert floats(r["o"]) == [4.0, 2.0, 1.0, 1.0]
    assert floats(r["o2"]) == [0.5, 0.75, 13.0, 11.0]
    assert floats(r["o3"]) == [1.0, 0.0, 1.0, 5.0]
    assert [x.v for x in r["b"].c] == [0x3f800000, 0x80000000, 1, 0]
    assert floats(r["o4"]) == [0.0, float(F(3) / F(5)), float(F(4) / F(5)), 1.5]


def test_arithmetic_is_binary32_without_contraction():
    r = run("""
        float a = 16777216.0; float b = a + 1.0;               // 2^24 + 1 is not a float
        float c = 0.1 * 3.0;                                   // rounded product of rounded literals
        float d = 1.0 / 3.0;
        float e = 1.0e-5;                                      // the reference's EPS
        o = vec4(b - a, c, d, e);
    """)
    assert floats(r["o"]) == [0.0, float(F(0.1) * F(3.0)), float(F(1.0) / F(3.0)), float(F(1e-5))]


def test_defines_and_constant_globals():
    r = run("o = vec4(TWO * 2.0, K[1].y, float(N), EPS);",
            decls="#define TWO 2.0\n#define EPS 1e-5\nconst vec2 K[] = vec2[](vec2(1, 2), vec2(3, 4));\nconst uint N = 7u;")
    assert floats(r["o"]) == [4.0, 4.0, 7.0, float(F(1e-5))]


def test_texture_filtering():
    tex = np.zeros((2, 4, 4), np.float32)
    tex[0, :, 0] = [0.0, 1.0, 2.0, 3.0]; tex[1, :, 0] = [10.0, 11.0, 12.0, 13.0]
    lin, near, rep = G.Sampler(tex, True), G.Sampler(tex, False), G.Sampler(tex, True, repeat=True)
    d = "uniform sampler2D t;"
    at = lambda s, x, y: float(run("o = texture(t, vec2(%r, %r));" % (x, y), decls=d, uniforms={"t": s})["o"].c[0])
    assert at(lin, 0.125, 0.25) == 0.0 and at(lin, 0.375, 0.25) == 1.0 and at(lin, 0.875, 0.75) == 13.0        # texel centres
    assert at(lin, 0.25, 0.25) == 0.5 and at(lin, 0.375, 0.5) == 6.0                                             # half-way: the mean
    assert at(lin, -3.0, 0.25) == 0.0 and at(lin, 7.0, 2.0) == 13.0                                              # CLAMP_TO_EDGE
    assert at(near, 0.26, 0.1) == 1.0 and at(near, 0.999, 0.6) == 13.0 and at(near, 1.5, -1.0) == 3.0            # NEAREST
    assert at(rep, 0.0, 0.25) == 1.5 and at(rep, 1.125, 0.25) == 0.0                                             # REPEAT: texel 3 and texel 0 meet at s = 0
    vol = np.zeros((2, 2, 2, 4), np.float32); vol[1, 1, 1, 0] = 8.0
    v = run("o = texture(t, vec3(0.5, 0.5, 0.5));", decls="uniform sampler3D t;", uniforms={"t": G.Sampler(vol, True)})["o"]
    assert float(v.c[0]) == 1.0                                                                                  # the mean of eight corners
    f = run("o = texelFetch(t, ivec2(2, 1), 0);", decls=d, uniforms={"t": near})["o"]
    assert float(f.c[0]) == 12.0


def test_errors_are_errors():
    with pytest.raises(G.GlslError):
        run("float a = 1.0 + 1u;")                             # no implicit conversions in GLSL ES
    with pytest.raises(G.GlslError):
        run("o = vec4(undeclared);")
    with pytest.raises(G.GlslError):
        run("vec3 v = vec3(1.0, 2.0);")                        # too few components
    with pytest.raises(G.GlslError):
        run("o = vec4(f(1));", decls="float f(float x) { return x; }")      # no overload for int
