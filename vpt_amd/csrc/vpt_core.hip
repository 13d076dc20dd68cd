// vpt_core.hip — context, volume (upload + re-layout), renderer life cycle, tile classification, options, read-back, counters, probes
// (C-ABI in include/vpt.h; the other translation units are listed in vpt_internal.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-gpu-flush-denormals-to-zero -fPIC (see vpt_amd/csrc/Makefile)
#include <chrono>
#include "vpt_internal.h"
#include "vpt_kernels_layout.h"
#include "vpt_srgb_lut.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
char *vpt_error_buffer(void) { return g_err; }
int fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char *vpt_last_error(void) { return g_err; }
extern "C" const char *vpt_version(void) { return "vpt-mi355x 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_device_count(int *count) {
    if (!count) return fail(VPT_ERR_INVALID, "count is null");
    HIP_TRY(hipGetDeviceCount(count));
    return VPT_OK;
}
extern "C" int vpt_context_create(int device, vpt_context **out) {
    if (!out) return fail(VPT_ERR_INVALID, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(VPT_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    vpt_context *c = new vpt_context();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(VPT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->owns_stream = true;
    *out = c;
    return VPT_OK;
}
extern "C" int vpt_context_create_on_stream(int device, void *hip_stream, vpt_context **out) {
    if (!out) return fail(VPT_ERR_INVALID, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(VPT_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    vpt_context *c = new vpt_context();
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->owns_stream = false;
    *out = c;
    return VPT_OK;
}
extern "C" int vpt_context_destroy(vpt_context *c) {
    if (!c) return VPT_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->owns_stream) hipStreamDestroy(c->stream);
    delete c;
    return VPT_OK;
}
extern "C" int vpt_context_synchronize(vpt_context *c) {
    if (!c) return fail(VPT_ERR_INVALID, "context is null");
    HIP_TRY(hipSetDevice(c->device));
    for (vpt_renderer *r : c->renderers) VPT_TRY(join_side(r));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// volume — Volume.js:31-78
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_volume_create(vpt_context *c, int w, int h, int d, int format, vpt_volume **out) {
    if (!c || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (format != VPT_FORMAT_R8 && format != VPT_FORMAT_RG8 && format != VPT_FORMAT_R32F && format != VPT_FORMAT_RG32F) return fail(VPT_ERR_UNSUPPORTED, "Unknown volume datatype: %d", format);  // Volume.js:103
    if (w < 1 || h < 1 || d < 1 || w > 4096 || h > 4096 || d > 4096)
        return fail(VPT_ERR_INVALID, "volume dimensions %dx%dx%d out of range [1,4096]", w, h, d);
    HIP_TRY(hipSetDevice(c->device));
    vpt_volume *v = new vpt_volume();
    memset(v, 0, sizeof(*v));
    v->ctx = c; v->nx = w; v->ny = h; v->nz = d;
    v->channels = (format == VPT_FORMAT_RG8 || format == VPT_FORMAT_RG32F) ? 2 : 1;
    v->f32 = format == VPT_FORMAT_R32F || format == VPT_FORMAT_RG32F;
    v->vox_bytes = v->channels * (v->f32 ? 4 : 1);
    // RG8: 256-byte slots (R brick at +0, G brick at +128); R32F: 512-byte slots; RG32F: 1024-byte slots (G brick at +512)
    const int slot_shift = (v->f32 ? 9 : 7) + (v->channels == 2 ? 1 : 0);
    const uint64_t eb = v->f32 ? 4 : 1;                  // bytes per texel channel
    v->filter = VPT_FILTER_LINEAR;                       // Volume.js:53-54
    int nbx = (w + 3) / 4, nby = (h + 3) / 4, nbz = (d + 3) / 4;
    // Z-order over the bricks with exactly as many bits per axis as the axis needs: the low bits of x, y, z interleave
    // (x lowest, as the classic Morton code), and once an axis runs out of bits the longer axes continue alone — a cube
    // gets the classic code, a 4096 x 2 x 3 volume 2^10 slots instead of 2^30.  code(bx,by,bz) = CX[bx] | CY[by] | CZ[bz].
    int nbits[3] = { 0, 0, 0 };
    { int nb[3] = { nbx, nby, nbz }; for (int ax = 0; ax < 3; ax++) while ((1 << nbits[ax]) < nb[ax]) nbits[ax]++; }
    int bitpos[3][16]; int total_bits = 0;
    for (int level = 0; level < 16; level++)
        for (int ax = 0; ax < 3; ax++) if (level < nbits[ax]) bitpos[ax][level] = total_bits++;
    auto axis_code = [&](int ax, uint32_t b) { uint64_t c = 0; for (int k = 0; k < nbits[ax]; k++) c |= (uint64_t)((b >> k) & 1u) << bitpos[ax][k]; return c; };
    size_t max_slot = (size_t)(axis_code(0, (uint32_t)nbx - 1) | axis_code(1, (uint32_t)nby - 1) | axis_code(2, (uint32_t)nbz - 1));
    v->brick_bytes = (max_slot + 1) << slot_shift;
    hipError_t e = hipMalloc(&v->linear, (size_t)w * h * d * v->vox_bytes);
    if (e == hipSuccess) e = hipMalloc(&v->bricks, v->brick_bytes + 64);   // +64: the 8-byte tap windows end <= byte 125+7
    if (e == hipSuccess) {
        // boundary atlas: six face images (axis x: ny x nz cells, y: nx x nz, z: nx x ny; low side, high side) with one common
        // power-of-two row pitch and one common size, one dword (byte volumes) or one float4 (float volumes) per cell and channel
        int pitch = 1, shift = 0;
        while (pitch < std::max(w, h)) { pitch <<= 1; shift++; }
        v->atlas_shift = (uint32_t)shift;
        v->atlas_face = (uint32_t)pitch * (uint32_t)std::max(h, d);
        v->atlas_dwords = 6 * (size_t)v->atlas_face * (size_t)v->channels * (v->f32 ? 4 : 1);
        e = hipMalloc(&v->atlas, v->atlas_dwords * 4);
        v->atlas_ok = true;
        if (e == hipSuccess && v->f32) e = hipMalloc(&v->atlas_flag, sizeof(uint32_t));
    }
    if (e != hipSuccess) {
        if (v->atlas) hipFree(v->atlas);
        if (v->bricks) hipFree(v->bricks);
        if (v->linear) hipFree(v->linear);
        delete v;
        return fail(VPT_ERR_HIP, "hipMalloc volume %dx%dx%d: %s", w, h, d, hipGetErrorString(e));
    }
    HIP_TRY(hipMemsetAsync(v->linear, 0, (size_t)w * h * d * v->vox_bytes, c->stream));   // texStorage3D zero-initialises
    {   // offset tables: off(x,y,z) = TX[x] + TY[y] + TZ[z]
        std::vector<uint64_t> t64((size_t)w + h + d);
        std::vector<uint32_t> t32(t64.size());
        for (int i = 0; i < w; i++) t64[i] = (axis_code(0, (uint32_t)i >> 2) << slot_shift) + (uint64_t)(i & 3) * eb;
        for (int i = 0; i < h; i++) t64[(size_t)w + i] = (axis_code(1, (uint32_t)i >> 2) << slot_shift) + (uint64_t)(i & 3) * 5 * eb;
        for (int i = 0; i < d; i++) t64[(size_t)w + h + i] = (axis_code(2, (uint32_t)i >> 2) << slot_shift) + (uint64_t)(i & 3) * 25 * eb;
        for (size_t i = 0; i < t64.size(); i++) t32[i] = (uint32_t)t64[i];
        for (size_t i = 0; i < t64.size(); i++) t64[i] >>= slot_shift;     // the brick's Morton code alone (WIDE variant)
        std::vector<uint32_t> tc(t64.size());
        for (size_t i = 0; i < t64.size(); i++) tc[i] = (uint32_t)t64[i];
        v->wide = v->brick_bytes > 0xffffffffull;
        HIP_TRY(hipMalloc(&v->tab32, t32.size() * 4));
        HIP_TRY(hipMalloc(&v->tabc, tc.size() * 4));
        HIP_TRY(hipMemcpy(v->tab32, t32.data(), t32.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(v->tabc, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
    }
    if (v->channels == 1 && !v->f32) {
        // column records: 2-D Z-order over (x, y) with as many bits per axis as the axis needs, nz records of 4 bytes per column
        int cb[2] = { 0, 0 };
        while ((1 << cb[0]) < w) cb[0]++;
        while ((1 << cb[1]) < h) cb[1]++;
        int cpos[2][16]; int ctotal = 0;
        for (int level = 0; level < 16; level++)
            for (int ax = 0; ax < 2; ax++) if (level < cb[ax]) cpos[ax][level] = ctotal++;
        auto col_code = [&](int ax, uint32_t i) { uint64_t c = 0; for (int k = 0; k < cb[ax]; k++) c |= (uint64_t)((i >> k) & 1u) << cpos[ax][k]; return c; };
        const uint64_t ncols = (col_code(0, (uint32_t)w - 1) | col_code(1, (uint32_t)h - 1)) + 1, col_bytes = 4ull * (uint64_t)d;
        v->rec_bytes = (size_t)(ncols * col_bytes);
        v->rec_wide = ncols * col_bytes > 0x100000000ull;          // (the largest offset used is rec_bytes - 4)
        std::vector<uint32_t> r32((size_t)w + h), rc((size_t)w + h);
        for (int i = 0; i < w; i++) { rc[i] = (uint32_t)col_code(0, (uint32_t)i); r32[i] = (uint32_t)(col_code(0, (uint32_t)i) * col_bytes); }
        for (int i = 0; i < h; i++) { rc[(size_t)w + i] = (uint32_t)col_code(1, (uint32_t)i); r32[(size_t)w + i] = (uint32_t)(col_code(1, (uint32_t)i) * col_bytes); }
        HIP_TRY(hipMalloc(&v->rtab32, r32.size() * 4));
        HIP_TRY(hipMalloc(&v->rtabc, rc.size() * 4));
        HIP_TRY(hipMemcpy(v->rtab32, r32.data(), r32.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(v->rtabc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice));
    }
    v->dirty = true;
    *out = v;
    return VPT_OK;
}
// The column records (vpt_device.h record_addr) of a one-channel byte volume, (re)built from the linear storage when blocks have been
// uploaded since the last build.  Allocated on first use: only the MCM renderer samples them (4 bytes per voxel).
int volume_records(vpt_volume *v) {
    if (!v || v->channels != 1 || v->f32) return fail(VPT_ERR_INVALID, "column records exist for one-channel byte volumes");
    VPT_TRY(vpt_volume_finalize(v));
    if (v->rec_valid) return VPT_OK;
    vpt_context *c = v->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (!v->records) HIP_TRY(hipMalloc(&v->records, v->rec_bytes + 64));      // + the dword behind the last column's last record
    for (vpt_renderer *r : c->renderers) if (r->vol == v) VPT_TRY(join_side(r));   // passes in flight read the old records
    hipLaunchKernelGGL(k_build_records, dim3((unsigned)((v->nx + 63) / 64), (unsigned)v->ny, (unsigned)((v->nz + 15) / 16)), dim3(256), 0, c->stream,
                       v->linear, v->records, v->nx, v->ny, v->nz, v->rec_wide ? v->rtabc : v->rtab32, v->rec_wide ? 1 : 0);
    HIP_TRY(hipGetLastError());
    v->rec_valid = true;
    for (vpt_renderer *r : c->renderers) if (r->vol == v) r->main_dirty = true;    // side streams must see the build
    return VPT_OK;
}
static int volume_upload(vpt_volume *v, int x, int y, int z, int w, int h, int d, const void *data, size_t nbytes, bool on_device) {
    if (!v || !data) return fail(VPT_ERR_INVALID, "null argument");
    if (w < 1 || h < 1 || d < 1 || x < 0 || y < 0 || z < 0 || x + w > v->nx || y + h > v->ny || z + d > v->nz)
        return fail(VPT_ERR_INVALID, "block (%d,%d,%d)+(%d,%d,%d) outside volume %dx%dx%d", x, y, z, w, h, d, v->nx, v->ny, v->nz);
    size_t need = (size_t)w * h * d * v->vox_bytes;
    if (nbytes < need) return fail(VPT_ERR_INVALID, "block data too short: %zu < %zu", nbytes, need);
    vpt_context *c = v->ctx;
    HIP_TRY(hipSetDevice(c->device));
    for (vpt_renderer *r : c->renderers) if (r->vol == v) VPT_TRY(join_side(r));
    bool full_xy = (x == 0 && y == 0 && w == v->nx && h == v->ny);
    if (full_xy) {   // contiguous run of z-slices (RAWReader.js:47-63 produces exactly these)
        uint8_t *dst = v->linear + (size_t)z * v->nx * v->ny * v->vox_bytes;
        HIP_TRY(hipMemcpyAsync(dst, data, need, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    } else {
        const uint8_t *src = (const uint8_t *)data;
        if (!on_device) {
            if (v->staging_bytes < need) {
                if (v->staging) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(v->staging)); v->staging = nullptr; }
                HIP_TRY(hipMalloc(&v->staging, need));
                v->staging_bytes = need;
            }
            HIP_TRY(hipMemcpyAsync(v->staging, data, need, hipMemcpyHostToDevice, c->stream));
            src = v->staging;
        }
        int grid = (int)((need / v->vox_bytes + 255) / 256); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(k_blit_block, dim3(grid), dim3(256), 0, c->stream, v->linear, v->nx, v->ny, src, x, y, z, w, h, d, v->vox_bytes);
        HIP_TRY(hipGetLastError());
    }
    if (!on_device) HIP_TRY(hipStreamSynchronize(c->stream));   // host buffer may be released by the caller
    v->dirty = true; v->any_upload = true; v->rec_valid = false;
    return VPT_OK;
}
extern "C" int vpt_volume_upload_block(vpt_volume *v, int x, int y, int z, int w, int h, int d, const void *data, size_t nbytes) {
    return volume_upload(v, x, y, z, w, h, d, data, nbytes, false);
}
extern "C" int vpt_volume_upload_block_device(vpt_volume *v, int x, int y, int z, int w, int h, int d, const void *data, size_t nbytes) {
    return volume_upload(v, x, y, z, w, h, d, data, nbytes, true);
}
extern "C" int vpt_volume_finalize(vpt_volume *v) {
    if (!v) return fail(VPT_ERR_INVALID, "volume is null");
    if (!v->dirty) return VPT_OK;
    vpt_context *c = v->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int nbx = (v->nx + 3) / 4, nby = (v->ny + 3) / 4, nbz = (v->nz + 3) / 4;
    if (nby > 65535 || nbz > 65535) return fail(VPT_ERR_UNSUPPORTED, "too many bricks");
    const int strips = (nbx + VPT_BRICKIFY_RUN - 1) / VPT_BRICKIFY_RUN;
    // one-channel volumes with dword-aligned rows go through the LDS-staged kernel (dword loads and stores)
    int fast = (v->channels == 1 && v->nx % 4 == 0) ? strips : 0;
    if (v->f32) {
        hipLaunchKernelGGL(k_brickify_f32, dim3((unsigned)strips, (unsigned)nby, (unsigned)nbz), dim3(128), 0, c->stream, (const float *)v->linear, (float *)v->bricks, v->nx, v->ny, v->nz, v->channels, v->tabc);
        fast = strips;                                    // nothing left for the byte kernels
    } else if (fast > 0)
        hipLaunchKernelGGL(k_brickify_strip, dim3((unsigned)fast, (unsigned)((nby + VPT_BRICKIFY_ROWS - 1) / VPT_BRICKIFY_ROWS), (unsigned)((nbz + VPT_BRICKIFY_ROWS - 1) / VPT_BRICKIFY_ROWS)), dim3(256), 0, c->stream, v->linear, v->bricks, v->nx, v->ny, v->nz, v->tabc);
    if (fast < strips)
        hipLaunchKernelGGL(k_brickify, dim3((unsigned)(strips - fast), (unsigned)nby, (unsigned)nbz), dim3(128), 0, c->stream, v->linear, v->bricks, v->nx, v->ny, v->nz, v->channels, v->tabc, fast * VPT_BRICKIFY_RUN);
    if (v->atlas) {
        size_t cells = (size_t)v->ny * v->nz + (size_t)v->nx * v->nz + (size_t)v->nx * v->ny;
        if (v->f32) {
            hipLaunchKernelGGL(k_build_atlas<float>, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream, (const float *)v->linear, (void *)v->atlas,
                               v->nx, v->ny, v->nz, v->channels, v->atlas_face, v->atlas_shift);
            // (float texels: the atlas stands for the bricks only while every texel is finite and differences cannot overflow — one scan, one
            // host wait per upload of a float volume)
            uint32_t bad = 0;
            HIP_TRY(hipMemsetAsync(v->atlas_flag, 0, sizeof(uint32_t), c->stream));
            hipLaunchKernelGGL(k_scan_finite, dim3(2048), dim3(256), 0, c->stream, (const float *)v->linear, (size_t)v->nx * v->ny * v->nz * v->channels, v->atlas_flag);
            HIP_TRY(hipMemcpyAsync(&bad, v->atlas_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            v->atlas_ok = bad == 0;
        } else {
            hipLaunchKernelGGL(k_build_atlas<uint8_t>, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)v->linear, (void *)v->atlas,
                               v->nx, v->ny, v->nz, v->channels, v->atlas_face, v->atlas_shift);
        }
    }
    HIP_TRY(hipGetLastError());
    v->dirty = false;
    return VPT_OK;
}
extern "C" int vpt_volume_set_filter(vpt_volume *v, int filter) {
    if (!v) return fail(VPT_ERR_INVALID, "volume is null");
    v->filter = (filter == VPT_FILTER_LINEAR) ? VPT_FILTER_LINEAR : VPT_FILTER_NEAREST;   // Volume.js:121
    return VPT_OK;
}
extern "C" int vpt_volume_set_wide_tables(vpt_volume *v, int wide) {
    if (!v) return fail(VPT_ERR_INVALID, "volume is null");
    if (!wide && v->brick_bytes > 0xffffffffull) return fail(VPT_ERR_INVALID, "bricked layout exceeds 4 GiB: 64-bit offset tables are required");
    v->wide = wide != 0;
    return VPT_OK;
}
extern "C" int vpt_volume_bricked_bytes(vpt_volume *v, uint64_t *n) {
    if (!v || !n) return fail(VPT_ERR_INVALID, "null argument");
    *n = v->brick_bytes;
    return VPT_OK;
}
static void renderers_unbind(vpt_context *c, vpt_volume *v) {
    for (vpt_renderer *r : c->renderers) if (r->vol == v) r->vol = nullptr;
}
extern "C" int vpt_volume_destroy(vpt_volume *v) {
    if (!v) return VPT_OK;
    hipSetDevice(v->ctx->device);
    for (vpt_renderer *r : v->ctx->renderers) if (r->vol == v) join_side(r);
    hipStreamSynchronize(v->ctx->stream);
    renderers_unbind(v->ctx, v);              // a renderer still bound to it reports "no ready volume" instead of reading freed memory
    if (v->linear) hipFree(v->linear);
    if (v->bricks) hipFree(v->bricks);
    if (v->atlas) hipFree(v->atlas);
    if (v->atlas_flag) hipFree(v->atlas_flag);
    if (v->records) hipFree(v->records);
    if (v->rtab32) hipFree(v->rtab32);
    if (v->rtabc) hipFree(v->rtabc);
    if (v->staging) hipFree(v->staging);
    if (v->tab32) hipFree(v->tab32);
    if (v->tabc) hipFree(v->tabc);
    delete v;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// renderer
// ---------------------------------------------------------------------------------------------
static void renderer_free_buffers(vpt_renderer *r) {
    if (r->frame) hipFree(r->frame);
    if (r->acc) hipFree(r->acc);
    for (int i = 0; i < 4; i++) if (r->st[i]) hipFree(r->st[i]);
    if (r->render) hipFree(r->render);
    if (r->scratch) hipFree(r->scratch);
    if (r->ndc_x) hipFree(r->ndc_x);
    if (r->ndc_y) hipFree(r->ndc_y);
    r->ndc_x = r->ndc_y = nullptr;
    r->frame = r->acc = nullptr; r->render = nullptr; r->scratch = nullptr; r->scratch_bytes = 0;
    for (int i = 0; i < 4; i++) r->st[i] = nullptr;
}
// _rebuildBuffers: AbstractRenderer.js:78-92 (+ the per-renderer buffer specs)
static int renderer_alloc_buffers(vpt_renderer *r) {
    vpt_context *c = r->ctx;
    if (r->play_graph) { hipStreamSynchronize(c->stream); play_graph_free(r->play_graph); r->play_graph = nullptr; }
    r->render_target = nullptr; r->target_is_callers = false;   // an external target was sized for the old geometry
    r->tm_owner = nullptr; r->tm_valid = false; r->tm_mode = 0;  // a fused tone mapper's output was sized for it too: it re-arms itself
    if (r->frame_ring) { hipFree(r->frame_ring); r->frame_ring = nullptr; } r->ring_frames = 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    renderer_free_buffers(r);
    r->cls.valid = false; r->cls.stale = false;            // new geometry, zeroed state: classes come back with the next reset
    r->cls.built = false;
    r->cls.passes = 0; r->cls.fused_passes = 0; r->cls.reset_seen = false; r->cls.n_complete = 0;   // (zeroed buffers are not a reset: nothing is skipped before one)
    r->dos_cur = 0; r->dos_rect_valid = false;
    int nblocks = (r->H + r->R - 1) / r->R;                 // row blocks in the image
    int mine = (nblocks - r->g + r->G - 1) / r->G;          // blocks b with b % G == g
    int max_blocks = (nblocks + r->G - 1) / r->G;           // every rank pads to this (equal-size gather)
    if (mine < 0) mine = 0;
    r->local_h = max_blocks * r->R;
    if (r->G == 1) r->local_h = r->H;
    r->tiles_x = (r->W + VPT_TILE - 1) / VPT_TILE;
    r->tiles_y = (r->local_h + VPT_TILE - 1) / VPT_TILE;
    r->ntiles = r->tiles_x * r->tiles_y;
    r->npix_padded = (size_t)r->ntiles * VPT_BLOCK;
    uint64_t valid = 0;
    for (int l = 0; l < r->local_h; l++) {
        int lb = l / r->R; int j = (lb * r->G + r->g) * r->R + (l - lb * r->R);
        if (j < r->H) valid += (uint64_t)r->W;
    }
    r->valid_pixels = valid;
    size_t fe = frame_elem(r->kind);
    if (fe) {
        HIP_TRY(hipMalloc(&r->frame, r->npix_padded * fe));
        HIP_TRY(hipMalloc(&r->acc, r->npix_padded * fe));
        HIP_TRY(hipMemsetAsync(r->frame, 0, r->npix_padded * fe, c->stream));
        HIP_TRY(hipMemsetAsync(r->acc, 0, r->npix_padded * fe, c->stream));
    } else {
        for (int i = 0; i < 4; i++) {
            // MCM: position (0) and transmittance (2) are 12-byte texels — their fourth float is a constant 0 in the reference's
            // attachments (MCMRenderer.glsl:168,170) and is not stored; DOS keeps float4 / float arrays in the same slots
            size_t texel = (r->kind == VPT_RENDERER_MCM && (i == 0 || i == 2)) ? 3 * sizeof(float) : sizeof(float4);
            HIP_TRY(hipMalloc(&r->st[i], r->npix_padded * texel));
            HIP_TRY(hipMemsetAsync(r->st[i], 0, r->npix_padded * texel, c->stream));
        }
    }
    {   // pixel-centre NDC: fl(fl((2i+1)/W) - 1), the exact per-pixel expression of the contract
        std::vector<float> nx((size_t)r->W), ny((size_t)r->H);
        for (int i = 0; i < r->W; i++) nx[i] = (float)(2 * i + 1) / (float)r->W - 1.0f;
        for (int j = 0; j < r->H; j++) ny[j] = (float)(2 * j + 1) / (float)r->H - 1.0f;
        HIP_TRY(hipMalloc(&r->ndc_x, nx.size() * sizeof(float)));
        HIP_TRY(hipMalloc(&r->ndc_y, ny.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(r->ndc_x, nx.data(), nx.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(r->ndc_y, ny.data(), ny.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    size_t rb = (size_t)r->W * r->local_h * sizeof(uint2);
    HIP_TRY(hipMalloc(&r->render, rb));
    HIP_TRY(hipMemsetAsync(r->render, 0, rb, c->stream));
    return VPT_OK;
}
static int upload_table(vpt_context *c, float4 **dst, const std::vector<float4> &host) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (*dst) { HIP_TRY(hipFree(*dst)); *dst = nullptr; }
    HIP_TRY(hipMalloc(dst, host.size() * sizeof(float4)));
    HIP_TRY(hipMemcpy(*dst, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice));
    return VPT_OK;
}
extern "C" int vpt_renderer_set_transfer_function(vpt_renderer *r, const uint8_t *rgba, int w, int h) {
    if (!r || !rgba) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    if (w < 1 || h < 1 || w > 2048 || h > 4096) return fail(VPT_ERR_INVALID, "transfer function size %dx%d out of range", w, h);
    // SRGB8_ALPHA8: rgb decoded before filtering, alpha linear (AbstractRenderer.js:36,99-104)
    std::vector<float4> t((size_t)w * h);
    for (size_t i = 0; i < t.size(); i++)
        t[i] = make_float4(VPT_SRGB_TO_LINEAR[rgba[4 * i]], VPT_SRGB_TO_LINEAR[rgba[4 * i + 1]],
                           VPT_SRGB_TO_LINEAR[rgba[4 * i + 2]], (float)rgba[4 * i + 3] / 255.0f);
    VPT_TRY(upload_table(r->ctx, &r->tf, t));
    r->tf_w = w; r->tf_h = h;
    return VPT_OK;
}
extern "C" int vpt_renderer_set_environment(vpt_renderer *r, const uint8_t *rgba, int w, int h) {
    if (!r || !rgba) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    if (w < 1 || h < 1 || w > 16384 || h > 16384) return fail(VPT_ERR_INVALID, "environment size %dx%d out of range", w, h);
    std::vector<float4> t((size_t)w * h);
    for (size_t i = 0; i < t.size(); i++)
        t[i] = make_float4((float)rgba[4 * i] / 255.0f, (float)rgba[4 * i + 1] / 255.0f,
                           (float)rgba[4 * i + 2] / 255.0f, (float)rgba[4 * i + 3] / 255.0f);
    VPT_TRY(upload_table(r->ctx, &r->env, t));
    r->env_w = w; r->env_h = h; r->env_const = t[0];
    r->env_opaque = true;
    for (size_t i = 0; i < t.size(); i++) if (rgba[4 * i + 3] != 255) { r->env_opaque = false; break; }
    r->cls.poisoned = true;                                  // (MCS: the fixed points of the ray-missing pixels move with the environment)
    return VPT_OK;
}
// VPT_OPTION_SPLIT_STREAMS as the library sets it itself (round 4: the measured best form is the default, not an option a caller has to know).
// MCM: the HIT | MISS kernels of the tile classes need two streams (1080p headline frame 80 us against 108 for the general kernel on one).
// MIP, EAM, Depth: three ranges on three streams (EAM 256^3 60.0 -> 48.4 us, MIP 50.5 -> 41.3, Depth 55.5 -> 46.7; two: 50.2 / 41.9 / 47.8).
// ISO: two (512^3: 47.0 -> 41.3 us, 256^3: 41.6 -> 42.2; three: 43.6 / 46.2 — its 7-sample shading of the few surface pixels does not balance).
// MCS, LAO: two (MCS 512^3 1080p 16.1 -> 13.8 us, extinction 50: 158 -> 124, extinction 200: 435 -> 358, 1024^3: 19.4 -> 16.6; LAO 2.11 -> 1.73 ms;
// three or four streams are no better and less stable from box to box).  DOS cannot split.  VPT_DEFAULT_SPLIT=1 in the environment: one stream everywhere.
static int default_split(int kind) {
    static const bool one = []() { const char *e = getenv("VPT_DEFAULT_SPLIT"); return e && e[0] == '1' && !e[1]; }();
    if (one) return 1;
    switch (kind) {
        case VPT_RENDERER_MCM: return 2;
        case VPT_RENDERER_MIP: case VPT_RENDERER_EAM: case VPT_RENDERER_DEPTH: return 3;
        case VPT_RENDERER_ISO: case VPT_RENDERER_MCS: case VPT_RENDERER_LAO: return 2;
        default: return 1;
    }
}
extern "C" int vpt_renderer_create(vpt_context *c, int kind, int width, int height, vpt_renderer **out) {
    if (!c || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (kind < VPT_RENDERER_MIP || kind > VPT_RENDERER_DOS) return fail(VPT_ERR_INVALID, "No suitable class");  // RendererFactory.js:21
    if (width < 1 || height < 1 || width > 32768 || height > 32768) return fail(VPT_ERR_INVALID, "resolution %dx%d out of range", width, height);
    HIP_TRY(hipSetDevice(c->device));
    vpt_renderer *r = new vpt_renderer();
    r->ctx = c; r->kind = kind; r->W = width; r->H = height;
    c->renderers.push_back(r);
    r->G = 1; r->g = 0; r->R = 8;
    r->vol = nullptr; r->tf = nullptr; r->env = nullptr;
    r->frame = r->acc = nullptr; r->render = nullptr; r->scratch = nullptr; r->scratch_bytes = 0;
    for (int i = 0; i < 4; i++) r->st[i] = nullptr;
    r->samples = nullptr; r->samples_host = 0; r->profiling = false; r->events_used = 0; r->profile_every = 1; r->profile_seq = 0;
    r->side_events_used = 0; r->timed_now = false;
    r->ndc_x = r->ndc_y = nullptr;
    r->frame_table = nullptr; r->frame_staging = nullptr; r->frame_counter = nullptr; r->frames_played = 0;
    r->warmed = false; r->play_graph = nullptr;
    r->fast_math = 0; r->boundary_atlas = 1; r->column_records = 2;
    r->frame_ring = nullptr; r->ring_frames = 0; r->split = default_split(kind); r->split_auto = true; r->target_is_callers = false; r->no_split = false; r->bucket_call = false; r->last_ranges = 1; r->stop_events = nullptr; r->stop_used = false; r->ev_fork = nullptr; for (int i = 0; i < VPT_MAX_SPLIT - 1; i++) { r->side[i] = nullptr; r->ev_join[i] = nullptr; } r->side_busy = false; r->main_dirty = true; r->mcm_persistent = 0; r->work_counter = nullptr; r->mcs_persistent = false;   // measured slower than k_mcs at every extinction tried
    r->render_target = nullptr;
    memset(&r->cls, 0, sizeof(r->cls)); r->cls.enabled = true; r->last_layout = 0; { const char *e = getenv("VPT_HIT_KERNEL_FORM"); r->hit_form = (e && (e[0] == '1' || e[0] == '2') && !e[1]) ? e[0] - '0' : 0; } r->bucket_kernel = false; r->bucket_launches = 0;
    r->tm_owner = nullptr; r->tm_valid = false; r->tm_table = nullptr; r->tm_out = nullptr; r->tm_mode = 0;
    r->lao = LaoParams{ 1, 0.69f, 1, 0.05f, 1, 0.54f, 10, 0.19f, 1.0f, { 2.0f, 12.0f, 3.0f } };
    int rc = renderer_alloc_buffers(r);
    if (rc == VPT_OK) {
        hipError_t e = hipMalloc(&r->samples, COUNTER_BYTES);
        if (e == hipSuccess) e = hipMemsetAsync(r->samples, 0, COUNTER_BYTES, c->stream);
        if (e != hipSuccess) rc = fail(VPT_ERR_HIP, "hipMalloc counter: %s", hipGetErrorString(e));
    }
    static const uint8_t default_tf[8] = { 255, 0, 0, 0, 255, 0, 0, 255 };   // AbstractRenderer.js:31-44
    static const uint8_t default_env[4] = { 255, 255, 255, 255 };            // RenderingContext.js:90-101
    if (rc == VPT_OK) rc = vpt_renderer_set_transfer_function(r, default_tf, 2, 1);
    if (rc == VPT_OK) rc = vpt_renderer_set_environment(r, default_env, 1, 1);
    if (rc != VPT_OK) { vpt_renderer_destroy(r); return rc; }
    *out = r;
    return VPT_OK;
}
extern "C" int vpt_renderer_destroy(vpt_renderer *r) {
    if (!r) return VPT_OK;
    join_side(r);
    hipSetDevice(r->ctx->device);
    hipStreamSynchronize(r->ctx->stream);
    tonemappers_unbind(r->ctx, r);            // a tone mapper still bound to this renderer falls back to the white placeholder
    for (size_t i = 0; i < r->ctx->renderers.size(); i++)
        if (r->ctx->renderers[i] == r) { r->ctx->renderers.erase(r->ctx->renderers.begin() + (long)i); break; }
    renderer_free_buffers(r);                 // renderer-owned buffers only; volume is NOT owned (Volume.js:17-22)
    if (r->tf) hipFree(r->tf);
    if (r->env) hipFree(r->env);
    if (r->samples) hipFree(r->samples);
    if (r->dos_samples) hipFree(r->dos_samples);
    if (r->work_counter) hipFree(r->work_counter);
    if (r->cls.list) hipFree(r->cls.list);
    if (r->cls.violations) hipFree(r->cls.violations);
    for (int i = 0; i < 2; i++) { if (r->cls.staging[i]) hipHostFree(r->cls.staging[i]); if (r->cls.staged[i]) hipEventDestroy(r->cls.staged[i]); }
    if (r->frame_ring) hipFree(r->frame_ring);
    if (r->frame_table) hipFree(r->frame_table);
    if (r->frame_staging) hipHostFree(r->frame_staging);
    if (r->frame_counter) hipFree(r->frame_counter);
    for (int i = 0; i < VPT_MAX_SPLIT - 1; i++) if (r->side[i]) { hipStreamDestroy(r->side[i]); hipEventDestroy(r->ev_join[i]); }
    if (r->ev_fork) hipEventDestroy(r->ev_fork);
    if (r->play_graph) play_graph_free(r->play_graph);
    for (auto &ev : r->events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (auto &ev : r->side_events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    delete r;
    return VPT_OK;
}
extern "C" int vpt_renderer_set_shard(vpt_renderer *r, int rank, int world, int rows_per_block) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (world < 1 || rank < 0 || rank >= world || rows_per_block < 1) return fail(VPT_ERR_INVALID, "bad shard %d/%d rows %d", rank, world, rows_per_block);
    if (r->kind == VPT_RENDERER_DOS && world > 1)
        return fail(VPT_ERR_UNSUPPORTED, "the DOS renderer does not shard: every slice reads its neighbours' occlusion across rows");
    r->G = world; r->g = rank; r->R = rows_per_block;
    return renderer_alloc_buffers(r);
}
extern "C" int vpt_renderer_local_rows(vpt_renderer *r, int *rows) {
    if (!r || !rows) return fail(VPT_ERR_INVALID, "null argument");
    *rows = r->local_h;
    return VPT_OK;
}
extern "C" int vpt_renderer_global_row(vpt_renderer *r, int l, int *j) {
    if (!r || !j) return fail(VPT_ERR_INVALID, "null argument");
    if (l < 0 || l >= r->local_h) return fail(VPT_ERR_INVALID, "local row %d out of range", l);
    int lb = l / r->R; int g = (lb * r->G + r->g) * r->R + (l - lb * r->R);
    *j = (g < r->H) ? g : -1;
    return VPT_OK;
}
extern "C" int vpt_renderer_resize(vpt_renderer *r, int width, int height) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (width < 1 || height < 1 || width > 32768 || height > 32768) return fail(VPT_ERR_INVALID, "resolution %dx%d out of range", width, height);
    if (width == r->W && height == r->H) return VPT_OK;   // AbstractRenderer.js:107
    r->W = width; r->H = height;
    return renderer_alloc_buffers(r);
}
extern "C" int vpt_renderer_set_volume(vpt_renderer *r, vpt_volume *v) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    if (v && v->ctx != r->ctx) return fail(VPT_ERR_INVALID, "volume belongs to another context");
    r->vol = v;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// tile classes: which 16x16 tiles can no camera ray of theirs take into the cube?  (host only, double precision)
// ---------------------------------------------------------------------------------------------
// inverse of a column-major float matrix by Gauss-Jordan with partial pivoting; out[row][col]; false: singular
bool invert_matrix(const float *m, double out[4][4]) {
    double a[4][8];
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) { a[row][col] = (double)m[col * 4 + row]; a[row][4 + col] = row == col ? 1.0 : 0.0; }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int row = col + 1; row < 4; row++) if (fabs(a[row][col]) > fabs(a[piv][col])) piv = row;
        if (!(fabs(a[piv][col]) > 1e-300)) return false;
        if (piv != col) for (int k = 0; k < 8; k++) std::swap(a[piv][k], a[col][k]);
        double inv = 1.0 / a[col][col];
        for (int k = 0; k < 8; k++) a[col][k] *= inv;
        for (int row = 0; row < 4; row++) if (row != col) { double f = a[row][col]; for (int k = 0; k < 8; k++) a[row][k] -= f * a[col][k]; }
    }
    for (int row = 0; row < 4; row++) for (int col = 0; col < 4; col++) out[row][col] = a[row][4 + col];
    return true;
}
// A camera ray of pixel (i, j) — unprojectRand with blur == 0 (mixins/unprojectRand.glsl:3-24) — joins the NDC points
// (x_i, y_j, -1) and (x_i + ax, y_j + ay, +1), |ax| <= 1/W, |ay| <= 1/H.  The unprojection is projective, so the world-space
// line is the image of that NDC line, whose point at depth z lies within (1/W, 1/H) * |z + 1| / 2 of (x_i, y_j).  The cube
// [0,1]^3 (taken as [-VPT_CLASS_EPS, 1 + VPT_CLASS_EPS]^3 against the kernels' fp32 rounding) with all eight corners in front
// of the eye plane (clip w > 0) maps onto the convex hull of its projected corners, at depths z in [zmin, zmax].  Hence: if the
// rectangle of a tile's pixel centres, widened by that drift at the cube's depths plus one pixel, does not touch the hull of the
// cube's projection, no ray of the tile — and no photon travelling along one — can be inside the cube: the tile is a MISS tile.
// Exact arithmetic is not needed, only conservatism: every doubt (a corner at or behind the eye plane, a singular matrix, wild
// depths) makes every tile a HIT tile, which is always correct.  classes[ty * tiles_x + tx] = 1 for MISS.  Rows are the LOCAL rows
// of shard (g of G, R rows per block): a tile whose 16 local rows are several runs of global rows is tested run by run.
#ifndef VPT_CLASS_EPS
#define VPT_CLASS_EPS 0.001
#endif
#define VPT_CLASS_FAR 1000.0
static void classify_tiles(int W, int H, int local_h, int G, int g, int R, const float *mvp_inverse, std::vector<uint8_t> &classes, int *ptx, int *pty) {
    const int tiles_x = (W + VPT_TILE - 1) / VPT_TILE, tiles_y = (local_h + VPT_TILE - 1) / VPT_TILE;
    *ptx = tiles_x; *pty = tiles_y;
    classes.assign((size_t)tiles_x * tiles_y, 0);
    double M[4][4];
    for (int k = 0; k < 16; k++) if (!(fabsf(mvp_inverse[k]) < 1e30f)) return;       // NaN / inf / absurd entries
    if (!invert_matrix(mvp_inverse, M)) return;
    // VPT_CLASS_EPS is an ABSOLUTE margin against the kernels' fp32 evaluation of from + t * direction, whose rounding grows with |from|
    // (about |from| * 2^-23 per operation): with the near-plane points beyond ~1e3 units (an orthographic-like or very distant camera) it
    // no longer covers it — every tile HIT
    for (int c = 0; c < 4; c++) {
        const double x = (c & 1) ? 1.0 : -1.0, y = (c & 2) ? 1.0 : -1.0;
        double q[4];
        for (int row = 0; row < 4; row++) q[row] = (double)mvp_inverse[row] * x + (double)mvp_inverse[4 + row] * y - (double)mvp_inverse[8 + row] + (double)mvp_inverse[12 + row];
        if (!(fabs(q[3]) > 1e-30)) return;
        for (int k = 0; k < 3; k++) if (!(fabs(q[k] / q[3]) < VPT_CLASS_FAR)) return;
    }
    double px[8], py[8], zmin = 1e300, zmax = -1e300;
    for (int c = 0; c < 8; c++) {
        const double e = VPT_CLASS_EPS;
        double p[4] = { (c & 1) ? 1.0 + e : -e, (c & 2) ? 1.0 + e : -e, (c & 4) ? 1.0 + e : -e, 1.0 }, q[4];
        for (int row = 0; row < 4; row++) q[row] = M[row][0] * p[0] + M[row][1] * p[1] + M[row][2] * p[2] + M[row][3] * p[3];
        if (!(q[3] > 1e-3)) return;                              // at or behind the eye plane
        px[c] = q[0] / q[3]; py[c] = q[1] / q[3];
        const double z = q[2] / q[3];
        if (!(fabs(px[c]) < 1e6) || !(fabs(py[c]) < 1e6) || !(fabs(z) < 1e3)) return;
        zmin = std::min(zmin, z); zmax = std::max(zmax, z);
    }
    // convex hull (monotone chain, counter-clockwise)
    int order[8]; for (int c = 0; c < 8; c++) order[c] = c;
    std::sort(order, order + 8, [&](int a, int b) { return px[a] < px[b] || (px[a] == px[b] && py[a] < py[b]); });
    auto cross = [&](double ax, double ay, double bx, double by, double cx, double cy) { return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax); };
    double hx[18], hy[18]; int nh = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int start = nh;
        for (int k = 0; k < 8; k++) {
            const int c = pass == 0 ? order[k] : order[7 - k];
            while (nh - start >= 2 && cross(hx[nh - 2], hy[nh - 2], hx[nh - 1], hy[nh - 1], px[c], py[c]) <= 0.0) nh--;
            hx[nh] = px[c]; hy[nh] = py[c]; nh++;
        }
        nh--;                                                    // the last point of a chain opens the next one
    }
    if (nh < 3) return;                                          // degenerate projection
    double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
    for (int k = 0; k < nh; k++) { bx0 = std::min(bx0, hx[k]); bx1 = std::max(bx1, hx[k]); by0 = std::min(by0, hy[k]); by1 = std::max(by1, hy[k]); }
    const double drift = 1.01 * std::max(fabs(zmin + 1.0), fabs(zmax + 1.0)) * 0.5;
    const double mx = drift / W + 2.0 / W, my = drift / H + 2.0 / H;      // the jitter's drift at the cube's depths + one pixel
    auto rect_misses = [&](double x0, double x1, double y0, double y1) {
        if (x1 < bx0 || x0 > bx1 || y1 < by0 || y0 > by1) return true;
        for (int k = 0; k < nh; k++) {                           // a hull edge with the whole rectangle strictly on its outer side
            const int k1 = (k + 1) % nh;
            const double ex = hx[k1] - hx[k], ey = hy[k1] - hy[k];
            const double tol = -1e-12 * (fabs(ex) + fabs(ey) + 1.0);
            if (cross(hx[k], hy[k], hx[k1], hy[k1], x0, y0) < tol && cross(hx[k], hy[k], hx[k1], hy[k1], x1, y0) < tol &&
                cross(hx[k], hy[k], hx[k1], hy[k1], x0, y1) < tol && cross(hx[k], hy[k], hx[k1], hy[k1], x1, y1) < tol) return true;
        }
        return false;
    };
    for (int ty = 0; ty < tiles_y; ty++) {
        // runs of consecutive global rows among the tile row's local rows (rows past the image are padding: no pixels)
        int run0[16], run1[16], nruns = 0;
        for (int l = ty * VPT_TILE; l < std::min((ty + 1) * VPT_TILE, local_h); l++) {
            int j = l;
            if (G > 1) { int lb = l / R; j = (lb * G + g) * R + (l - lb * R); }
            if (j >= H) continue;
            if (nruns && run1[nruns - 1] + 1 == j) run1[nruns - 1] = j;
            else { run0[nruns] = run1[nruns] = j; nruns++; }
        }
        for (int tx = 0; tx < tiles_x; tx++) {
            const int i0 = tx * VPT_TILE, i1 = std::min(i0 + VPT_TILE - 1, W - 1);
            const double x0 = (2.0 * i0 + 1.0) / W - 1.0 - mx, x1 = (2.0 * i1 + 1.0) / W - 1.0 + mx;
            bool miss = true;
            for (int k = 0; k < nruns && miss; k++)
                miss = rect_misses(x0, x1, (2.0 * run0[k] + 1.0) / H - 1.0 - my, (2.0 * run1[k] + 1.0) / H - 1.0 + my);
            classes[(size_t)ty * tiles_x + tx] = miss ? 1 : 0;
        }
    }
}
// (extension, host only: no GPU is touched) the classification as tests/test_tile_classes.py checks it against brute force
extern "C" int vpt_classify_tiles(int width, int height, int rank, int world, int rows_per_block, const float *mvp_inverse,
                                  uint8_t *classes, size_t nclasses, int *tiles_x, int *tiles_y) {
    if (!mvp_inverse || !tiles_x || !tiles_y) return fail(VPT_ERR_INVALID, "null argument");
    if (width < 1 || height < 1 || width > 32768 || height > 32768 || world < 1 || rank < 0 || rank >= world || rows_per_block < 1)
        return fail(VPT_ERR_INVALID, "bad geometry");
    int local_h = height;
    if (world > 1) { int nblocks = (height + rows_per_block - 1) / rows_per_block; local_h = ((nblocks + world - 1) / world) * rows_per_block; }
    std::vector<uint8_t> cls;
    classify_tiles(width, height, local_h, world, rank, rows_per_block, mvp_inverse, cls, tiles_x, tiles_y);
    if (classes) {
        if (nclasses < cls.size()) return fail(VPT_ERR_INVALID, "classes buffer too small: %zu < %zu", nclasses, cls.size());
        memcpy(classes, cls.data(), cls.size());
    }
    return VPT_OK;
}

int make_args(vpt_renderer *r, const vpt_uniforms *u, bool need_volume, PassArgs *a) {
    memset(a, 0, sizeof(*a));
    a->pm.W = r->W; a->pm.H = r->H; a->pm.local_h = r->local_h;
    a->pm.tiles_x = r->tiles_x; a->pm.ntiles = r->ntiles;
    a->pm.G = r->G; a->pm.g = r->g; a->pm.R = r->R;
    a->pm.rshift = -1;
    a->pm.ty0 = 0;
    for (int sft = 0; sft < 16; sft++) if ((1 << sft) == r->R) a->pm.rshift = sft;
    a->pm.ndc_x = r->ndc_x; a->pm.ndc_y = r->ndc_y;
    if (need_volume) {
        if (!r->vol || !r->vol->any_upload) return fail(VPT_ERR_NO_VOLUME, "renderer has no ready volume");
        VPT_TRY(vpt_volume_finalize(r->vol));
        vpt_volume *v = r->vol;
        a->vol.bricks = v->bricks; a->vol.nx = v->nx; a->vol.ny = v->ny; a->vol.nz = v->nz;
        a->vol.fnx = (float)v->nx; a->vol.fny = (float)v->ny; a->vol.fnz = (float)v->nz;
        a->vol.hx = (float)(v->nx - 1); a->vol.hy = (float)(v->ny - 1); a->vol.hz = (float)(v->nz - 1);
        a->vol.tab32 = v->tab32; a->vol.tabc = v->tabc;
        a->vol.filter = v->filter;
        a->vol.channels = v->channels; a->vol.slot_shift = (v->f32 ? 9u : 7u) + (v->channels == 2 ? 1u : 0u);
        a->vol.elem_shift = v->f32 ? 2u : 0u;
        a->vol.atlas = (r->boundary_atlas && v->atlas_ok) ? v->atlas : nullptr;
        a->vol.atlas_face = v->atlas_face; a->vol.atlas_shift = v->atlas_shift;
        if (renderer_uses_records(r)) {
            VPT_TRY(volume_records(v));
            a->vol.records = v->records; a->vol.rtab32 = v->rtab32; a->vol.rtabc = v->rtabc; a->vol.rec_col_bytes = 4u * (uint32_t)v->nz;
        }
    }
    a->env.texels = r->env; a->env.w = r->env_w; a->env.h = r->env_h; a->env.constant = r->env_const;
    a->tf = r->tf; a->tf_w = r->tf_w; a->tf_h = r->tf_h; a->tf_fw = (float)r->tf_w; a->tf_hi = (float)(r->tf_w - 1);
    if (u) {
        memcpy(a->mvp_inv.m, u->mvp_inverse, sizeof(float) * 16);
        a->seed = u->rand_seed; a->offset = u->offset; a->step = u->step_size;
        a->extinction = u->extinction; a->anisotropy = u->anisotropy;
        a->inv_extinction = 1.0f / u->extinction;      // -log(u)/rate is evaluated as -log(u) * (1/rate)
        a->max_bounces = u->max_bounces; a->steps = u->steps;
        a->light = f3{ u->light_direction[0], u->light_direction[1], u->light_direction[2] };
        a->mix = u->mix; a->blur = u->blur;
        a->isovalue = u->isovalue; a->gradient_step = u->gradient_step; a->threshold = u->threshold;
    }
    if (r->kind == VPT_RENDERER_LAO) a->lao = r->lao;
    a->inv_w = (float)(1.0 / (double)r->W);     // gl.uniform2f(uInverseResolution, 1/res, 1/res): MCMRenderer.js:91,155
    a->inv_h = (float)(1.0 / (double)r->H);
    a->frame = r->frame; a->acc = r->acc;
    a->st0 = r->st[0]; a->st1 = r->st[1]; a->st2 = r->st[2]; a->st3 = r->st[3];
    if (r->kind == VPT_RENDERER_DOS) {            // colour: st[0], in place; occlusion: in = the latest of st[2|3], out = the other
        a->st1 = nullptr; a->st2 = r->st[2 + r->dos_cur]; a->st3 = r->st[3 - r->dos_cur];
    }
    a->render = r->render_target ? r->render_target : r->render;
    if (r->tm_owner && r->tm_mode && !r->render_target) {
        a->tm_table = r->tm_table;
    }
    a->samples = r->samples;
    return VPT_OK;
}
// ---- streams that really run side by side ----------------------------------------------------------------------------------
// HIP gives a stream one of a few hardware queues (four by default) and does not say which: two streams on one queue execute their
// kernels one after the other.  Measured: the three tile-row ranges of an EAM frame 51.9 us on three queues, 83 us when the process had
// created one or two other streams first; the gather pipeline's hand-off 1.5 or 6 us.  So a stream that has to
// overlap others is PICKED: candidates are created one by one and each is tried against the streams it must overlap until one passes (at
// most 8; the rejected ones are destroyed afterwards, the first candidate stands if none passes, e.g. under a profiler that serialises
// dispatches).  The test is a rendezvous, not a timing (round 4; rounds 1-3 compared two 100 us wall-clock measurements, which eight ranks
// probing at once on one host can blur): kernel A on the one stream waits for a flag that kernel B on the other stream raises.  On two
// queues B runs beside A and A sees the flag within microseconds; on one queue B cannot start before A has ended, and A gives up after
// VPT_PROBE_TICKS of the wall clock — every wave leaves by the flag, the clock or an iteration cap.  The flag lives in pinned host memory
// (coherent for both kernels whichever XCD they run on).
// VPT_STREAM_PROBE=0 in the environment: take the first candidate, as rounds 1-2 did.
#define VPT_PROBE_TICKS 20000ull                                      // 200 us of the 100 MHz wall clock
__global__ void k_probe_wait(uint32_t *flag, uint32_t *seen, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    uint32_t it = 0, v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) == 0u && wall_clock64() - t0 < ticks && ++it < 2000000u) {}
    *seen = v;
}
__global__ void k_probe_raise(uint32_t *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
static bool streams_overlap(hipStream_t a, hipStream_t b) {
    static uint32_t *words = nullptr;                                // [0] flag, [1] seen: pinned, kept for the life of the process
    if (!words && hipHostMalloc((void **)&words, 2 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { words = nullptr; (void)hipGetLastError(); return true; }
    bool seen = false;
    for (int round = 0; round < 2 && !seen; round++) {               // (round 0 also pays for the code object's first use on these queues)
        hipStreamSynchronize(a); hipStreamSynchronize(b);
        words[0] = 0u; words[1] = 0u;
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, words, words + 1, VPT_PROBE_TICKS);
        hipLaunchKernelGGL(k_probe_raise, dim3(1), dim3(1), 0, b, words);
        hipStreamSynchronize(a); hipStreamSynchronize(b);
        seen = words[1] != 0u;
    }
    (void)hipGetLastError();
    return seen;
}
// a new non-blocking stream that overlaps every stream of `others` (null entries skipped)
hipError_t create_overlapping_stream(hipStream_t *out, const hipStream_t *others, int n_others) {
    static const bool probe = []() { const char *e = getenv("VPT_STREAM_PROBE"); return !(e && e[0] == '0'); }();
    hipStream_t tried[8]; int nt = 0; hipStream_t chosen = nullptr;
    while (nt < 8 && !chosen) {
        hipStream_t c;
        hipError_t e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
        if (e != hipSuccess) { if (nt == 0) return e; (void)hipGetLastError(); break; }
        tried[nt++] = c;
        bool ok = true;
        for (int i = 0; i < n_others && ok && probe; i++) if (others[i]) ok = streams_overlap(others[i], c);
        if (ok) chosen = c;
    }
    if (!chosen) chosen = tried[0];
    for (int i = 0; i < nt; i++) if (tried[i] != chosen) hipStreamDestroy(tried[i]);
    (void)hipGetLastError();
    *out = chosen;
    return hipSuccess;
}

// the side streams of VPT_OPTION_SPLIT_STREAMS = r->split, created (and picked: create_overlapping_stream) when the option is set or, for
// the library's default, by the first pass that wants them
int ensure_split_streams(vpt_renderer *r) {
    if (r->split < 2 || (r->side[r->split - 2] && r->ev_fork)) return VPT_OK;
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (!r->ev_fork) HIP_TRY(hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming));
    for (int i = 0; i + 1 < r->split; i++) if (!r->side[i]) {
        hipStream_t others[VPT_MAX_SPLIT] = { r->ctx->stream };      // the context's stream and the side streams there are
        for (int k = 0; k < VPT_MAX_SPLIT - 1; k++) others[1 + k] = r->side[k];
        HIP_TRY(hipStreamSynchronize(r->ctx->stream));
        HIP_TRY(create_overlapping_stream(&r->side[i], others, VPT_MAX_SPLIT));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_join[i], hipEventDisableTiming));
    }
    r->main_dirty = true;
    return VPT_OK;
}
// the side stream's work happens-before everything enqueued on the context's stream from here on
int join_side(vpt_renderer *r) {
    if (!r || !r->side_busy) return VPT_OK;
    for (int i = 0; i < VPT_MAX_SPLIT - 1; i++) if (r->side[i]) {
        HIP_TRY(hipEventRecord(r->ev_join[i], r->side[i]));
        HIP_TRY(hipStreamWaitEvent(r->ctx->stream, r->ev_join[i], 0));
    }
    r->side_busy = false; r->main_dirty = true;
    return VPT_OK;
}
// classifies the tiles for `mvp_inverse` (see classify_tiles) and puts the lists on the device: HIT tiles first, then MISS tiles.
// Host cost per call (1080p: 8 160 tiles): ~60 us of classification + one asynchronous 32 KB upload from pinned memory, no host wait;
// nothing at all when the lists on the device already describe this matrix and geometry.
int classes_build(vpt_renderer *r, const float *mvp_inverse) {
    TileClasses &c = r->cls;
    c.valid = false;
    const int geom[6] = { r->W, r->H, r->local_h, r->G, r->g, r->R };
    if (c.built && c.list && memcmp(c.built_mvp, mvp_inverse, sizeof(c.built_mvp)) == 0 && memcmp(c.built_geom, geom, sizeof(geom)) == 0) {
        memmove(c.mvp, mvp_inverse, sizeof(c.mvp));
        c.valid = true;                                         // (n_hit / n_miss / the device lists are those of the last build)
        return VPT_OK;
    }
    std::vector<uint8_t> cls; int tx, ty;
    classify_tiles(r->W, r->H, r->local_h, r->G, r->g, r->R, mvp_inverse, cls, &tx, &ty);
    if (tx != r->tiles_x || ty != r->tiles_y || tx > 0xffff || ty > 0xffff) return VPT_OK;
    const int n = (int)cls.size(), s = c.stage_next;
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (c.staging_capacity[s] < n) {
        if (c.staged[s]) HIP_TRY(hipEventSynchronize(c.staged[s]));
        if (c.staging[s]) { HIP_TRY(hipHostFree(c.staging[s])); c.staging[s] = nullptr; }
        HIP_TRY(hipHostMalloc((void **)&c.staging[s], (size_t)n * sizeof(uint32_t), hipHostMallocDefault));
        c.staging_capacity[s] = n;
    }
    if (!c.staged[s]) HIP_TRY(hipEventCreateWithFlags(&c.staged[s], hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(c.staged[s]));            // the copy out of this buffer two builds ago: long done (no wait in practice)
    uint32_t *list = c.staging[s];
    int nh = 0, nm = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int y = 0; y < ty; y++) for (int x = 0; x < tx; x++)
            if ((int)cls[(size_t)y * tx + x] == pass) { list[(size_t)nh + nm] = (uint32_t)x | ((uint32_t)y << 16); (pass ? nm : nh)++; }
    VPT_TRY(join_side(r));                                      // passes in flight read the old lists
    if (c.capacity < n) {
        HIP_TRY(hipStreamSynchronize(r->ctx->stream));
        if (c.list) { HIP_TRY(hipFree(c.list)); c.list = nullptr; }
        HIP_TRY(hipMalloc(&c.list, (size_t)n * sizeof(uint32_t)));
        c.capacity = n;
    }
    if (!c.violations) {       // [0]: VPT_OPTION_VERIFY_TILE_CLASSES
        HIP_TRY(hipMalloc(&c.violations, 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(c.violations, 0, 16 * sizeof(unsigned long long), r->ctx->stream));
    }
    // (the lists travel on the context's stream, behind the passes that read the old ones; the side streams pick them up at the next fork)
    HIP_TRY(hipMemcpyAsync(c.list, list, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, r->ctx->stream));
    HIP_TRY(hipEventRecord(c.staged[s], r->ctx->stream));
    c.stage_next = s ^ 1;
    r->main_dirty = true;
    c.n_hit = nh; c.n_miss = nm;
    memmove(c.mvp, mvp_inverse, sizeof(c.mvp));
    memcpy(c.built_mvp, mvp_inverse, sizeof(c.built_mvp)); memcpy(c.built_geom, geom, sizeof(geom)); c.built = true;
    c.valid = true;
    return VPT_OK;
}
// ---------------------------------------------------------------------------------------------
// read-back, counters, profiling
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_renderer_read(vpt_renderer *r, int buffer, void *dst, size_t nbytes) {
    if (!r || !dst) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    vpt_context *c = r->ctx;
    HIP_TRY(hipSetDevice(c->device));
    size_t npix = (size_t)r->W * r->local_h;
    if (buffer == VPT_BUFFER_RENDER) {
        if (nbytes < npix * 8) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, npix * 8);
        HIP_TRY(hipMemcpyAsync(dst, r->render_target ? r->render_target : r->render, npix * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return VPT_OK;
    }
    const void *src = nullptr; size_t elem = 0;
    if (r->kind == VPT_RENDERER_DOS) {            // row-major already: no tile order to undo
        if (buffer == VPT_BUFFER_ACCUM) { src = r->st[0]; elem = 16; }
        else if (buffer == VPT_BUFFER_DOS_OCCLUSION) { src = r->st[2 + r->dos_cur]; elem = 4; }
        else return fail(VPT_ERR_INVALID, "the DOS renderer holds VPT_BUFFER_ACCUM (colour) and VPT_BUFFER_DOS_OCCLUSION; its frame buffer is never written");
        if (nbytes < npix * elem) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, npix * elem);
        HIP_TRY(hipMemcpyAsync(dst, src, npix * elem, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return VPT_OK;
    }
    if (buffer == VPT_BUFFER_FRAME || buffer == VPT_BUFFER_ACCUM) {
        elem = frame_elem(r->kind);
        if (!elem) return fail(VPT_ERR_INVALID, "MCM has no frame/accumulation colour buffer; read the MCM state buffers");
        src = (buffer == VPT_BUFFER_FRAME) ? r->frame : r->acc;
    } else if (buffer >= VPT_BUFFER_MCM_POSITION && buffer <= VPT_BUFFER_MCM_RADIANCE) {
        if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_INVALID, "not an MCM renderer");
        if (buffer == VPT_BUFFER_MCM_POSITION || buffer == VPT_BUFFER_MCM_TRANSMITTANCE) VPT_TRY(mcm_materialize(r));
        elem = 16; src = r->st[buffer - VPT_BUFFER_MCM_POSITION];
    } else {
        return fail(VPT_ERR_INVALID, "unknown buffer %d", buffer);
    }
    if (nbytes < npix * elem) return fail(VPT_ERR_INVALID, "destination too small: %zu < %zu", nbytes, npix * elem);
    if (r->scratch_bytes < npix * elem) {
        if (r->scratch) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(r->scratch)); r->scratch = nullptr; }
        HIP_TRY(hipMalloc(&r->scratch, npix * elem));
        r->scratch_bytes = npix * elem;
    }
    PassArgs a;
    VPT_TRY(make_args(r, nullptr, false, &a));
    if (r->kind == VPT_RENDERER_MCM && (buffer == VPT_BUFFER_MCM_POSITION || buffer == VPT_BUFFER_MCM_TRANSMITTANCE))
        hipLaunchKernelGGL(k_detile_mcm3, tile_grid(r), dim3(VPT_BLOCK), 0, c->stream, a.pm, (const f3 *)src, (float4 *)r->scratch);
    else
    hipLaunchKernelGGL(k_detile, tile_grid(r), dim3(VPT_BLOCK), 0, c->stream, a.pm, (const uint8_t *)src, (uint8_t *)r->scratch, (int)elem);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst, r->scratch, npix * elem, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VPT_OK;
}
extern "C" int vpt_renderer_read_frame_slot(vpt_renderer *r, int slot, void *dst, size_t nbytes) {
    if (!r || !dst) return fail(VPT_ERR_INVALID, "null argument");
    if (!r->frame_ring || slot < 0 || slot >= r->ring_frames) return fail(VPT_ERR_INVALID, "frame slot %d: the last VPT_PLAY_FRAMES call wrote %d frames", slot, r->ring_frames);
    size_t need = (size_t)r->W * r->local_h * 8;
    if (nbytes < need) return fail(VPT_ERR_INVALID, "buffer too small: %zu < %zu", nbytes, need);
    VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, (const char *)r->frame_ring + (size_t)slot * need, need, hipMemcpyDeviceToHost, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    return VPT_OK;
}
extern "C" int vpt_renderer_frame_ring_device(vpt_renderer *r, void **ptr, size_t *slot_bytes) {
    if (!r || !ptr || !slot_bytes) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));                                   // (whoever gets the address works on the context's stream: the ring's writers come first)
    *ptr = r->frame_ring; *slot_bytes = (size_t)r->W * r->local_h * 8;
    return VPT_OK;
}
extern "C" int vpt_renderer_render_buffer_device(vpt_renderer *r, void **ptr, size_t *nbytes) {
    if (!r || !ptr || !nbytes) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));
    *ptr = r->render_target ? r->render_target : r->render; *nbytes = (size_t)r->W * r->local_h * 8;
    return VPT_OK;
}
extern "C" int vpt_renderer_bucket_launches(vpt_renderer *r, uint64_t *launches) {
    if (!r || !launches) return fail(VPT_ERR_INVALID, "null argument");
    *launches = r->bucket_launches;
    return VPT_OK;
}
extern "C" int vpt_renderer_join(vpt_renderer *r) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    HIP_TRY(hipSetDevice(r->ctx->device));
    return join_side(r);
}
extern "C" int vpt_renderer_set_render_target(vpt_renderer *r, void *ptr, size_t nbytes) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));                                   // (launches in flight carry their target in their arguments)
    size_t need = (size_t)r->W * r->local_h * 8;
    if (ptr && nbytes < need) return fail(VPT_ERR_INVALID, "render target too small: %zu < %zu", nbytes, need);
    r->render_target = (uint2 *)ptr; r->target_is_callers = ptr != nullptr;
    // what the library knows about this memory's texels (marcher_track: tiles it may skip) ends here: the caller may have used it in between
    for (int i = 0; i < r->cls.n_complete; i++)
        if (ptr && r->cls.complete[i] == ptr) { r->cls.complete[i] = r->cls.complete[--r->cls.n_complete]; break; }
    return VPT_OK;
}
extern "C" int vpt_renderer_set_lao_params(vpt_renderer *r, const struct vpt_lao_params *p) {
    if (!r || !p) return fail(VPT_ERR_INVALID, "null argument");
    if (r->kind != VPT_RENDERER_LAO) return fail(VPT_ERR_INVALID, "not an LAO renderer");
    if (p->num_lao_samples < 1 || p->num_lao_samples > 64 || p->num_shadow_samples < 1 || p->num_shadow_samples > 1024)
        return fail(VPT_ERR_INVALID, "sample counts out of range (LAO 1..64, shadows 1..1024)");
    if (!(p->lao_step_size >= 1.0f / 4096.0f)) return fail(VPT_ERR_INVALID, "LAO step size below 1/4096 (the occlusion march would not end)");
    static_assert(sizeof(LaoParams) == sizeof(vpt_lao_params), "parameter block layout");
    memcpy(&r->lao, p, sizeof(r->lao));
    return VPT_OK;
}
extern "C" int vpt_renderer_set_option(vpt_renderer *r, int option, int value) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    switch (option) {
        // (the persistent kernels walk every tile from the context's stream: ranges of earlier split passes must be in first)
        case VPT_OPTION_MCS_PERSISTENT: VPT_TRY(join_side(r)); r->mcs_persistent = value != 0; return VPT_OK;
        case VPT_OPTION_MCM_PERSISTENT: VPT_TRY(join_side(r)); r->mcm_persistent = value < 0 ? 0 : (value > 2 ? 2 : value); return VPT_OK;
        case VPT_OPTION_BOUNDARY_ATLAS: r->boundary_atlas = value != 0; return VPT_OK;
        case VPT_OPTION_SPLIT_STREAMS:
            if (r->kind == VPT_RENDERER_DOS) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_SPLIT_STREAMS: the DOS renderer's slices depend on each other across pixels");
            if (value < 1 || value > VPT_MAX_SPLIT) return fail(VPT_ERR_INVALID, "VPT_OPTION_SPLIT_STREAMS: 1 .. %d", VPT_MAX_SPLIT);
            VPT_TRY(join_side(r));
            r->split = value; r->split_auto = false;          // a count the caller asked for is taken as it is (vpt_internal.h split_for)
            return ensure_split_streams(r);
        case VPT_OPTION_FAST_MATH:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_FAST_MATH: only the MCM renderer has a fast-arithmetic variant");
            if ((value != 0) != (r->fast_math != 0)) VPT_TRY(mcm_materialize(r));   // MISS-tile positions in the arithmetic that produced the directions
            r->fast_math = value != 0; return VPT_OK;
        case VPT_OPTION_TILE_CLASSES:
            if (r->kind == VPT_RENDERER_DOS || r->kind == VPT_RENDERER_LAO) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_TILE_CLASSES: not an option of the DOS / LAO renderers");
            r->cls.enabled = value != 0; r->cls.one_stream = value == 2; return VPT_OK;
        case VPT_OPTION_COLUMN_RECORDS:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_COLUMN_RECORDS: an MCM option");
            VPT_TRY(join_side(r));
            if (value < 0 || value > 2) return fail(VPT_ERR_INVALID, "VPT_OPTION_COLUMN_RECORDS: 0 (bricks), 1 (records) or 2 (by volume size)");
            r->column_records = value; return VPT_OK;
        case VPT_OPTION_BUCKET_KERNEL:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_BUCKET_KERNEL: an MCM option");
            r->bucket_kernel = value != 0; return VPT_OK;
        case VPT_OPTION_VERIFY_TILE_CLASSES:
            if (r->kind != VPT_RENDERER_MCM) return fail(VPT_ERR_UNSUPPORTED, "VPT_OPTION_VERIFY_TILE_CLASSES: an MCM option");
            r->cls.verify = value != 0; return VPT_OK;
        default: return fail(VPT_ERR_INVALID, "unknown option %d", option);
    }
}
extern "C" int vpt_renderer_sample_count(vpt_renderer *r, uint64_t *count) {
    if (!r || !count) return fail(VPT_ERR_INVALID, "null argument");
    VPT_TRY(join_side(r));                                   // the ranges of a split pass count into the same slots
    HIP_TRY(hipSetDevice(r->ctx->device));
    unsigned long long slots[VPT_COUNTER_SLOTS * VPT_COUNTER_STRIDE];
    HIP_TRY(hipMemcpyAsync(slots, r->samples, COUNTER_BYTES, hipMemcpyDeviceToHost, r->ctx->stream));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    uint64_t dev = 0;
    for (int i = 0; i < VPT_COUNTER_SLOTS; i++) dev += slots[i * VPT_COUNTER_STRIDE];
    *count = dev + r->samples_host;
    return VPT_OK;
}
extern "C" int vpt_renderer_clear_sample_count(vpt_renderer *r) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    VPT_TRY(join_side(r));
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipMemsetAsync(r->samples, 0, COUNTER_BYTES, r->ctx->stream));
    r->samples_host = 0;
    return VPT_OK;
}
extern "C" int vpt_renderer_tile_classes(vpt_renderer *r, int *hit_tiles, int *miss_tiles, uint64_t *violations) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    if (hit_tiles) *hit_tiles = r->cls.valid ? r->cls.n_hit : r->ntiles;
    if (miss_tiles) *miss_tiles = r->cls.valid ? r->cls.n_miss : 0;
    if (violations) {
        *violations = 0;
        if (r->cls.violations) {
            VPT_TRY(join_side(r));
            HIP_TRY(hipSetDevice(r->ctx->device));
            unsigned long long v = 0;
            HIP_TRY(hipMemcpyAsync(&v, r->cls.violations, sizeof(v), hipMemcpyDeviceToHost, r->ctx->stream));
            HIP_TRY(hipStreamSynchronize(r->ctx->stream));
            *violations = v;
        }
    }
    return VPT_OK;
}
extern "C" int vpt_renderer_set_profiling(vpt_renderer *r, int enabled) {
    if (!r) return fail(VPT_ERR_INVALID, "renderer is null");
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    r->profiling = enabled != 0;
    r->profile_every = enabled > 1 ? enabled : 1;   // enabled = n > 1: every n-th launch only (events cost ~7 us per launch)
    r->profile_seq = 0;
    r->events_used = 0; r->side_events_used = 0;
    return VPT_OK;
}
extern "C" int vpt_renderer_profile_side(vpt_renderer *r, double *total_ms, uint32_t *launches) {
    if (!r || !total_ms || !launches) return fail(VPT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(r->ctx->device));
    VPT_TRY(join_side(r));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    double sum = 0.0;
    for (size_t i = 0; i < r->side_events_used; i++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r->side_events[i].first, r->side_events[i].second));
        sum += (double)ms;
    }
    *total_ms = sum; *launches = (uint32_t)r->side_events_used;
    return VPT_OK;
}
extern "C" int vpt_renderer_profile(vpt_renderer *r, double *total_ms, uint32_t *launches) {
    if (!r || !total_ms || !launches) return fail(VPT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(r->ctx->device));
    HIP_TRY(hipStreamSynchronize(r->ctx->stream));
    double sum = 0.0;
    for (size_t i = 0; i < r->events_used; i++) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, r->events[i].first, r->events[i].second));
        sum += (double)ms;
    }
    uint32_t n = 0;
    for (size_t i = 0; i < r->events_used; i++) n += r->event_launches[i];
    *total_ms = sum; *launches = n;
    return VPT_OK;
}

// ---------------------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------------------
extern "C" int vpt_probe_math(vpt_context *c, int which, const float *in, float *out, size_t n) {
    if (!c || !in || !out) return fail(VPT_ERR_INVALID, "null argument");
    if (which < 0 || which > VPT_PROBE_POW) return fail(VPT_ERR_INVALID, "unknown probe %d", which);
    if (n == 0) return VPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    size_t nin = (which == VPT_PROBE_ATAN2 || which == VPT_PROBE_MIN || which == VPT_PROBE_MAX || which == VPT_PROBE_POW) ? 2 * n : n;
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&din, nin * sizeof(float)));
    hipError_t e = hipMalloc(&dout, n * sizeof(float));
    if (e != hipSuccess) { hipFree(din); return fail(VPT_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpyAsync(din, in, nin * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, which, din, dout, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(din); hipFree(dout);
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "probe: %s", hipGetErrorString(e));
    return VPT_OK;
}
static int probe_sample(vpt_renderer *r, const float *xyz, float *rgba, size_t n, bool boundary) {
    if (!r || !xyz || !rgba) return fail(VPT_ERR_INVALID, "null argument");
    if (n == 0) return VPT_OK;
    vpt_context *c = r->ctx;
    HIP_TRY(hipSetDevice(c->device));
    PassArgs a;
    VPT_TRY(make_args(r, nullptr, true, &a));
    if (boundary && !a.vol.atlas) return fail(VPT_ERR_UNSUPPORTED, "the volume's boundary atlas is not in use (switched off, or a float volume with non-finite texels)");
    float *din = nullptr; float4 *dout = nullptr;
    HIP_TRY(hipMalloc(&din, 3 * n * sizeof(float)));
    hipError_t e = hipMalloc(&dout, n * sizeof(float4));
    if (e != hipSuccess) { hipFree(din); return fail(VPT_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpyAsync(din, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        dim3 grid((unsigned)((n + 255) / 256));
        switch (variant_of(r)) {
            case 0: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<0>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<0>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 1: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<1>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<1>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 2: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<2>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<2>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 3: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<3>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<3>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 8: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<8>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<8>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 9: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<9>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<9>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 10: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<10>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<10>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 11: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<11>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<11>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 32: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<32>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<32>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 33: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<33>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<33>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 34: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<34>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<34>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 35: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<35>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<35>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 40: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<40>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<40>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 41: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<41>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<41>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            case 42: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<42>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<42>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
            default: if (boundary) hipLaunchKernelGGL(k_probe_sample_boundary<43>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); else hipLaunchKernelGGL(k_probe_sample<43>, grid, dim3(VPT_BLOCK), lds_bytes(r), c->stream, a, din, dout, n); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgba, dout, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(din); hipFree(dout);
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "probe: %s", hipGetErrorString(e));
    return VPT_OK;
}

extern "C" int vpt_probe_sample(vpt_renderer *r, const float *xyz, float *rgba, size_t n) { return probe_sample(r, xyz, rgba, n, false); }
extern "C" int vpt_probe_sample_boundary(vpt_renderer *r, const float *xyz, float *rgba, size_t n) { return probe_sample(r, xyz, rgba, n, true); }

extern "C" int vpt_probe_stream_read(vpt_context *c, size_t nbytes, int iterations, double *gb_per_s) {
    if (!c || !gb_per_s) return fail(VPT_ERR_INVALID, "null argument");
    if (nbytes < (1u << 20) || iterations < 1) return fail(VPT_ERR_INVALID, "need at least 1 MiB and one iteration");
    HIP_TRY(hipSetDevice(c->device));
    size_t n16 = nbytes / 16;
    uint4 *buf = nullptr; uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&buf, n16 * 16);
    if (e == hipSuccess) e = hipMalloc(&sink, 4);
    if (e == hipSuccess) e = hipMemsetAsync(buf, 0, n16 * 16, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    if (e == hipSuccess) {
        dim3 grid(256 * 16);                                  // 16 workgroups per CU
        hipLaunchKernelGGL(k_stream_read, grid, dim3(VPT_BLOCK), 0, c->stream, buf, n16, sink);    // warm-up
        e = hipEventRecord(e0, c->stream);
        for (int i = 0; i < iterations && e == hipSuccess; i++) {
            hipLaunchKernelGGL(k_stream_read, grid, dim3(VPT_BLOCK), 0, c->stream, buf, n16, sink);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (buf) hipFree(buf);
    if (sink) hipFree(sink);
    if (e != hipSuccess) return fail(VPT_ERR_HIP, "stream probe: %s", hipGetErrorString(e));
    *gb_per_s = (double)(n16 * 16) * iterations / ((double)ms * 1e-3) / 1e9;
    return VPT_OK;
}
