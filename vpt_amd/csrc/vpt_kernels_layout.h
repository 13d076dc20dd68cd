// vpt_kernels_layout.h — layout conversion (upload, re-layout, read-back), the streaming-read probe and the test probes; included by vpt_core.hip only.
#pragma once
#include "vpt_kernels.h"


// =============================================================================================
// layout helpers
// =============================================================================================
// tile-order per-pixel buffer -> row-major local rows (read-back only); elem = bytes per pixel
__global__ void __launch_bounds__(VPT_BLOCK) k_detile(PixMap pm, const uint8_t *src, uint8_t *dst, int elem) {
    Pix p = map_pixel(pm);
    if (!(p.i < pm.W && p.l < pm.local_h)) return;
    const uint8_t *s = src + (size_t)p.k * elem;
    uint8_t *d = dst + ((size_t)p.l * pm.W + p.i) * elem;
    for (int b = 0; b < elem; b++) d[b] = s[b];
}

// the MCM position / transmittance arrays (12-byte texels, tile order) -> RGBA32F rows with w = 0, as the reference's attachments hold them
__global__ void __launch_bounds__(VPT_BLOCK) k_detile_mcm3(PixMap pm, const f3 *src, float4 *dst) {
    Pix p = map_pixel(pm);
    if (!(p.i < pm.W && p.l < pm.local_h)) return;
    f3 v = src[p.k];
    dst[(size_t)p.l * pm.W + p.i] = make_float4(v.x, v.y, v.z, 0.0f);
}

// texSubImage3D: contiguous block (bw x bh x bd, `ch` interleaved bytes per voxel) -> linear volume at (x0,y0,z0)
__global__ void k_blit_block(uint8_t *vol, int nx, int ny, const uint8_t *blk, int x0, int y0, int z0, int bw, int bh, int bd, int ch) {
    size_t n = (size_t)bw * bh * bd;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        int x = (int)(t % bw); size_t r = t / bw; int y = (int)(r % bh); int z = (int)(r / bh);
        size_t dst = (((size_t)(z0 + z) * ny + (y0 + y)) * nx + (x0 + x)) * ch;
        for (int c = 0; c < ch; c++) vol[dst + c] = blk[t * ch + c];
    }
}
// linear volume -> apron bricks in Morton order; one 128-thread workgroup per brick
// (3-D grid: a 1-D grid of 2048^3's 2^27 bricks x 128 threads exceeds HIP's 2^32 work-items per dimension)
// `ch` = 1 (R8: 128-byte slots) or 2 (RG8: 256-byte slots, the R brick at +0 and the G brick at +128)
// codes = the brick-code tables CX | CY | CZ (indexed by voxel coordinate): slot(bx,by,bz) = CX[4bx] + CY[4by] + CZ[4bz]
// a workgroup re-lays VPT_BRICKIFY_RUN consecutive bricks of a brick row (one brick per workgroup: 2^21 tiny workgroups for
// 512^3, 0.74 ms = 0.5 TB/s, bound by workgroup launches)
#define VPT_BRICKIFY_RUN 16
__global__ void __launch_bounds__(128) k_brickify(const uint8_t *lin, uint8_t *bricks, int nx, int ny, int nz, int ch, const uint32_t *codes, int first_brick) {
    const int by = (int)blockIdx.y, bz = (int)blockIdx.z;
    const int t = (int)threadIdx.x;
    const int nbx = (nx + VPT_BRICK - 1) / VPT_BRICK;
    const int lx = t % 5, ly = (t / 5) % 5, lz = t / 25;
    const int y = min(by * VPT_BRICK + ly, ny - 1), z = min(bz * VPT_BRICK + lz, nz - 1);
    const size_t row = ((size_t)z * ny + y) * nx;
    const uint32_t cyz = codes[nx + 4 * by] + codes[nx + ny + 4 * bz];
    const int bx0 = first_brick + (int)blockIdx.x * VPT_BRICKIFY_RUN;
    for (int c = 0; c < ch; c++) {
        uint8_t v[VPT_BRICKIFY_RUN];
#pragma unroll
        for (int u = 0; u < VPT_BRICKIFY_RUN; u++) {
            int bx = min(bx0 + u, nbx - 1);
            int x = min(bx * VPT_BRICK + lx, nx - 1);
            v[u] = (t < 125) ? lin[(row + x) * ch + c] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < VPT_BRICKIFY_RUN; u++) {
            int bx = bx0 + u;
            if (bx < nbx) {
                size_t slot = (size_t)(codes[4 * bx] + cyz) << (ch == 2 ? 8 : 7);
                bricks[slot + (size_t)c * 128 + t] = v[u];
            }
        }
    }
}

// The same re-layout for one-channel volumes whose rows are dword-aligned (nx % 4 == 0), staged through LDS: a workgroup
// takes a block of 16 x 4 x 4 bricks (64 x 16 x 16 voxels + the apron column, row and slice): 17 x 17 source rows of 17
// dwords, loaded as dwords with all of a thread's loads in flight before its first LDS write, then writes the 256 brick
// slots as 16-byte pieces in the order of the brick codes — with Z-order codes the block is four contiguous 8 KiB runs of the
// brick array (a wave instruction = eight whole consecutive slots).  No division in either loop.
#define VPT_BRICKIFY_ROWS 4       // brick rows (y) and brick layers (z) per workgroup of k_brickify_strip
__global__ void __launch_bounds__(256) k_brickify_strip(const uint8_t *lin, uint8_t *bricks, int nx, int ny, int nz, const uint32_t *codes) {
    constexpr int NR = VPT_BRICK * VPT_BRICKIFY_ROWS + 1;                             // 17 voxel rows / slices incl. the apron
    __shared__ uint32_t rows[NR * NR][17];
    const int by0 = (int)blockIdx.y * VPT_BRICKIFY_ROWS, bz0 = (int)blockIdx.z * VPT_BRICKIFY_ROWS, t = (int)threadIdx.x;
    const int nby = (ny + VPT_BRICK - 1) / VPT_BRICK, nbz = (nz + VPT_BRICK - 1) / VPT_BRICK;
    const int x0 = (int)blockIdx.x * (VPT_BRICK * VPT_BRICKIFY_RUN);                 // multiple of 64; nx % 4 == 0 guaranteed by the launch
    {   // thread t < 255 loads dword d = t % 17 of the rows t / 17 + 15 i, i = 0 .. 19 (row r = slice r / 17, voxel row r % 17)
        const int d = t % 17, r0 = t / 17;
        if (t < 255) {
            // past the row's end (the apron of the last brick column, the unused tail of a partial strip): voxel nx-1 replicated
            const bool inside = x0 + 4 * d < nx;
            uint32_t v[20];
            int ry = r0, zi = 0;
#pragma unroll
            for (int i = 0; i < 20; i++) {
                if (zi < NR) {
                    const int y = min(by0 * VPT_BRICK + ry, ny - 1), z = min(bz0 * VPT_BRICK + zi, nz - 1);
                    const uint8_t *row = lin + ((size_t)z * ny + y) * nx;
                    v[i] = inside ? *(const uint32_t *)(row + x0 + 4 * d) : (uint32_t)row[nx - 1] * 0x01010101u;
                }
                ry += 15; if (ry >= NR) { ry -= NR; zi++; }
            }
            int r = r0;
#pragma unroll
            for (int i = 0; i < 20; i++) {
                if (r < NR * NR) rows[r][d] = v[i];
                r += 15;
            }
        }
    }
    __syncthreads();
    const uint8_t *lb = (const uint8_t *)rows;
    const int w8 = t & 7, s = t >> 3;                        // 16-byte piece w8 of brick u = s + 32 it of the block
    // byte b = 16 w8 + k of a brick = voxel (lx, ly, lz), b = lx + 5 ly + 25 lz; 125..127 are padding
    const int b0 = 16 * w8, lz0 = b0 / 25, rem0 = b0 - 25 * lz0, ly0 = rem0 / 5, lx0 = rem0 - 5 * ly0;
#pragma unroll 2
    for (int it = 0; it < 8; it++) {
        // u in Z-order over the low two bits of (ux, uy, uz), then the high bits of ux: consecutive u = consecutive brick codes
        const int u = s + 32 * it;
        const int ux = (u & 1) | ((u >> 2) & 2) | ((u >> 4) & 12), uy = ((u >> 1) & 1) | ((u >> 3) & 2), uz = ((u >> 2) & 1) | ((u >> 4) & 2);
        if (x0 + 4 * ux >= nx || by0 + uy >= nby || bz0 + uz >= nbz) continue;
        const int base = ((VPT_BRICK * uz) * NR + VPT_BRICK * uy) * 68 + 4 * ux;
        uint32_t o[4] = { 0u, 0u, 0u, 0u };
        int lx = lx0, ly = ly0, lz = lz0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (b0 + k < 125) o[k >> 2] |= (uint32_t)lb[base + (lz * NR + ly) * 68 + lx] << (8 * (k & 3));
            if (++lx == 5) { lx = 0; if (++ly == 5) { ly = 0; lz++; } }
        }
        const size_t slot = (size_t)(codes[x0 + 4 * ux] + codes[nx + 4 * (by0 + uy)] + codes[nx + ny + 4 * (bz0 + uz)]) << 7;
        *(uint4 *)(bricks + slot + 16 * w8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// FLOAT volumes: linear floats -> 5^3-float apron bricks in 512-byte slots (slot = brick code << 9; two channels: 1024-byte slots,
// the G brick 512 bytes behind the R brick); VPT_BRICKIFY_RUN bricks of a brick row per workgroup, thread t < 125 carries local
// voxel t of each
__global__ void __launch_bounds__(128) k_brickify_f32(const float *lin, float *bricks, int nx, int ny, int nz, int ch, const uint32_t *codes) {
    const int by = (int)blockIdx.y, bz = (int)blockIdx.z, t = (int)threadIdx.x;
    if (t >= 125) return;
    const int nbx = (nx + VPT_BRICK - 1) / VPT_BRICK;
    const int lx = t % 5, ly = (t / 5) % 5, lz = t / 25;
    const int y = min(by * VPT_BRICK + ly, ny - 1), z = min(bz * VPT_BRICK + lz, nz - 1);
    const size_t row = ((size_t)z * ny + y) * nx;
    const uint32_t cyz = codes[nx + 4 * by] + codes[nx + ny + 4 * bz];
    const int shift = ch == 2 ? 8 : 7;                                          // floats per slot: 128 or 256
    for (int u = 0; u < VPT_BRICKIFY_RUN; u++) {
        int bx = (int)blockIdx.x * VPT_BRICKIFY_RUN + u;
        if (bx >= nbx) break;
        int x = min(bx * VPT_BRICK + lx, nx - 1);
        const size_t slot = (size_t)(codes[4 * bx] + cyz) << shift;
        for (int c = 0; c < ch; c++) bricks[slot + (size_t)c * 128 + t] = lin[(row + x) * ch + c];
    }
}

// column records (vpt_device.h record_addr): a workgroup takes 64 voxels of a row x 16 slices of one y, stages the rows y and
// y + 1 (clamped) in LDS and writes, per column, the 16 records of its z range as one contiguous 64-byte run (four lanes x four
// dwords).  offsets = RX | RY (nx + ny entries): the column's first record in bytes, or its Z-order code when `wide`
// (then byte offset = code * 4 nz).  Any size: indices clamped, partial tiles guarded.
__global__ void __launch_bounds__(256) k_build_records(const uint8_t *lin, uint8_t *records, int nx, int ny, int nz, const uint32_t *offsets, int wide) {
    __shared__ uint8_t tile[2][16][68];
    const int t = (int)threadIdx.x;
    const int x0 = (int)blockIdx.x * 64, y = (int)blockIdx.y, z0 = (int)blockIdx.z * 16;
    const int y1 = min(y + 1, ny - 1);
    for (int idx = t; idx < 2 * 16 * 65; idx += 256) {
        const int xi = idx % 65, rz = idx / 65, zi = rz & 15, rw = rz >> 4;
        const int x = min(x0 + xi, nx - 1), z = min(z0 + zi, nz - 1);
        tile[rw][zi][xi] = lin[((size_t)z * ny + (rw ? y1 : y)) * nx + x];
    }
    __syncthreads();
    const int xl = t >> 2, zq = (t & 3) * 4, x = x0 + xl;
    if (x >= nx) return;
    const uint32_t o = offsets[x] + offsets[nx + y];
    uint8_t *col = records + (wide ? (uint64_t)o * (uint64_t)(4 * nz) : (uint64_t)o);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int zi = zq + k, z = z0 + zi;
        if (z < nz)
            ((uint32_t *)col)[z] = (uint32_t)tile[0][zi][xl] | ((uint32_t)tile[0][zi][xl + 1] << 8) | ((uint32_t)tile[1][zi][xl] << 16) | ((uint32_t)tile[1][zi][xl + 1] << 24);
    }
}

// boundary atlas (vpt_device.h sample_volume_boundary): thread c of [0, cx + cy + cz) builds cell c of the low-side AND the
// high-side face of its axis from the linear volume.  Face x: cells (a, b) = (y, z); y: (x, z); z: (x, y); face f = 2 * axis +
// side at dword f * face, cell (a, b) at (b << shift) + a.
// T = uint8_t: one dword per cell (four bytes); T = float: one 16-byte cell (four floats).  `ch` interleaved channels in the linear volume:
// channel c's six face images start 6 * face cells behind channel c - 1's.
template <typename T>
__global__ void __launch_bounds__(256) k_build_atlas(const T *lin, void *atlas, int nx, int ny, int nz, int ch, uint32_t face, uint32_t shift) {
    size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t cx = (size_t)ny * nz, cy = (size_t)nx * nz, cz = (size_t)nx * ny;
    int axis, na, nb;
    if (c < cx) { axis = 0; na = ny; nb = nz; }
    else if (c < cx + cy) { axis = 1; c -= cx; na = nx; nb = nz; }
    else if (c < cx + cy + cz) { axis = 2; c -= cx + cy; na = nx; nb = ny; }
    else return;
    const int a = (int)(c % (size_t)na), b = (int)(c / (size_t)na);
    const int a1 = min(a + 1, na - 1), b1 = min(b + 1, nb - 1);
    const int nk = axis == 0 ? nx : (axis == 1 ? ny : nz);
    for (int chan = 0; chan < ch; chan++)
        for (int side = 0; side < 2; side++) {
            const int k = side ? nk - 1 : 0;
            auto vox = [&](int p, int q) -> T {
                int x = axis == 0 ? k : p, y = axis == 0 ? p : (axis == 1 ? k : q), z = axis == 2 ? k : q;
                return lin[(((size_t)z * ny + y) * nx + x) * ch + chan];
            };
            const size_t cell = ((size_t)chan * 6 + (size_t)(2 * axis + side)) * face + ((size_t)b << shift) + a;
            if (sizeof(T) == 1)
                ((uint32_t *)atlas)[cell] = (uint32_t)vox(a, b) | ((uint32_t)vox(a1, b) << 8) | ((uint32_t)vox(a, b1) << 16) | ((uint32_t)vox(a1, b1) << 24);
            else
                ((float4 *)atlas)[cell] = make_float4((float)vox(a, b), (float)vox(a1, b), (float)vox(a, b1), (float)vox(a1, b1));
        }
}
// float volumes: is every texel finite and small enough that a difference of two texels cannot overflow?  Only then is
// lerp(t, t', 0) = fma(0, t' - t, t) == t, which the boundary atlas (and with it the tile classes) relies on for float texels.
__global__ void __launch_bounds__(256) k_scan_finite(const float *lin, size_t n, uint32_t *bad) {
    uint32_t b = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b |= !(fabsf(lin[i]) < 1e37f);
    if (__builtin_amdgcn_ballot_w64(b != 0) && ((int)threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// streaming read: every lane pulls 16 B per iteration, grid-stride; the xor keeps the loads alive
typedef unsigned int vpt_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(VPT_BLOCK) void k_stream_read(const uint4 *src_, size_t n16, uint32_t *sink) {
    const vpt_u32x4 *src = (const vpt_u32x4 *)src_;
    size_t stride = (size_t)gridDim.x * VPT_BLOCK;
    vpt_u32x4 acc = { 0u, 0u, 0u, 0u };
    size_t i = (size_t)blockIdx.x * VPT_BLOCK + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        vpt_u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        vpt_u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= src[i];
    uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (v == 0x9E3779B9u) *sink = v;                      // practically never: the buffer is zero-filled
}

// probes (tests)
__global__ void k_probe_math(int which, const float *in, float *out, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float r = 0.0f, s, c;
    switch (which) {
        case 0: r = vpt_logf(in[t]); break;
        case 1: vpt_sincosf(in[t], s, c); r = s; break;
        case 2: vpt_sincosf(in[t], s, c); r = c; break;
        case 3: r = vpt_asinf(in[t]); break;
        case 4: r = vpt_atan2f(in[2 * t], in[2 * t + 1]); break;
        case 5: r = __uint_as_float(pcg(__float_as_uint(in[t]))); break;
        case 6: { uint32_t st = __float_as_uint(in[t]); r = random_uniform(st); } break;
        case 7: r = __uint_as_float((uint32_t)to_half_bits(in[t])); break;
        case 8: r = rcp_nr(in[t]); break;
        case 9: r = rsqrt_nr(in[t]); break;
        case 10: r = vmin(in[2 * t], in[2 * t + 1]); break;
        case 11: r = vmax(in[2 * t], in[2 * t + 1]); break;
        case 12: r = vpt_logf_uniform(in[t]); break;
        case 13: r = rcp_nrz(in[t]); break;
        case 14: r = sqrt_nr(in[t]); break;
        case 15: r = vpt_expf(in[t]); break;
        case 16: r = vpt_powf(in[2 * t], in[2 * t + 1]); break;
    }
    out[t] = r;
}
// the same sample through the boundary atlas (positions with a coordinate outside [0, 1]; others through the bricks): what k_mcm_miss and the
// out-of-cube lanes of k_mcm_integrate execute — must equal k_probe_sample bit for bit
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) k_probe_sample_boundary(PassArgs a, const float *xyz, float4 *out, size_t n) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 q = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
    const bool oob = (vmax(vmax(q.x, q.y), q.z) > 1.0f) || (vmin(vmin(q.x, q.y), q.z) < 0.0f);
    if (!oob) { out[i] = sample_volume_color<V>(a, t, q); return; }
    const f2 rg = sample_boundary_rg<V>(a.vol, q);
    out[i] = (V & VPT_V_RG) ? sample_tf2d(a.tf, a.tf_w, a.tf_h, rg.x, rg.y) : sample_tf(t.tf, a.tf_fw, a.tf_hi, rg.x);
}
template <int V>
__global__ void __launch_bounds__(VPT_BLOCK) k_probe_sample(PassArgs a, const float *xyz, float4 *out, size_t n) {
    extern __shared__ float4 lds_raw[];
    LdsTables t = stage_lds<(V & VPT_V_WIDE) != 0>(lds_raw, a);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sample_volume_color<V>(a, t, f3{ xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] });
}
