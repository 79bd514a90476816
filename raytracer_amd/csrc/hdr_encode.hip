// Radiance .hdr encoding of a float RGB frame on the GPU: the bytes stbi_write_hdr writes for the same image
// (src/third_party/stb/stb_image_write.h:639-803; every demo ends with it, e.g. src/demos/cornell/main.c:222), so that
// 4 bytes per pixel or fewer cross PCIe instead of 12 and the host does no per-pixel work.
//
// File layout (stb_image_write.h:761-785): a text header, then per scanline
//   width < 8 or width >= 32768: the RGBE pixels, 4 bytes each;
//   otherwise: the marker {2, 2, width >> 8, width & 255} and the four components (all R bytes, all G, all B, all E)
//   run-length coded one after the other: a stretch without three equal bytes in a row is copied in packets
//   {n, n bytes} of n <= 128, a run of >= 3 equal bytes becomes packets {128 + n, byte} of n <= 127 (:722-757).
//
// The writer's greedy loop cuts a component into maximal runs of >= 3 equal bytes and the stretches between them, and
// both kinds are cut into packets from their own start.  That is a function of each byte's neighbourhood and of two
// running positions (start of the current run, start of the current stretch), so a wave codes a component 64 bytes
// at a time: one ballot of "equal to the next byte" per block, shifted and combined with the masks of the
// neighbouring blocks, tells every byte whether it lies in a run; further ballots give the two start positions and
// the output offset of every byte (a byte emits 0, 1 or 2 bytes), and the last byte of each packet writes the
// packet's length.
//
// Three launches: hdr_rows_kernel (one workgroup per scanline: RGBE conversion into LDS planes, then one wave per
// component codes its plane into a staging slot and records the length), hdr_offsets_kernel (scan of the lengths,
// header, total size), hdr_gather_kernel (staging slots -> their place in the file).

#include "rthip_internal.h"

namespace {

constexpr int kBlock = 256;

struct HdrHead {
    uint8_t text[124];
    uint32_t len;
};

// (unsigned char)(float) as the reference's x86-64 build performs it: truncation to a 32-bit integer, low byte;
// out-of-range and NaN inputs give the low byte of INT_MIN, i.e. 0.
__device__ __forceinline__ uint32_t byte_of_float(float x) {
    const int32_t v = (x >= -2147483648.f && x < 2147483648.f) ? (int32_t)x : INT32_MIN;
    return (uint32_t)v & 0xffu;
}

// stbiw__linear_to_rgbe, stb_image_write.h:639-654; returns r | g << 8 | b << 16 | e << 24
__device__ __forceinline__ uint32_t rgbe_of(float r, float g, float b) {
    const float m12 = g > b ? g : b;
    const float maxcomp = r > m12 ? r : m12;
    if (maxcomp < 1e-32f) return 0u;
    const uint32_t bits = __float_as_uint(maxcomp);
    const uint32_t field = (bits >> 23) & 0xffu;
    // maxcomp >= 1e-32 is a normal number, +inf or NaN.  frexp of inf / NaN returns its argument with exponent 0,
    // which makes the scale NaN and every mantissa byte 0.
    if (field == 0xffu) return 128u << 24;
    const int exponent = (int)field - 126;                                     // maxcomp = mant * 2^exponent
    const float mant = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);  // in [0.5, 1)
    const float normalize = mant * 256.0f / maxcomp;
    return byte_of_float(r * normalize) | byte_of_float(g * normalize) << 8 | byte_of_float(b * normalize) << 16 |
           ((uint32_t)(exponent + 128) & 0xffu) << 24;
}

__global__ __launch_bounds__(kBlock) void hdr_flat_kernel(const float* __restrict__ rgb, uint64_t n_pixels,
                                                          uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_pixels) return;
    const uint32_t v = rgbe_of(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    out[4 * i] = (uint8_t)v;
    out[4 * i + 1] = (uint8_t)(v >> 8);
    out[4 * i + 2] = (uint8_t)(v >> 16);
    out[4 * i + 3] = (uint8_t)(v >> 24);
}

// One workgroup per scanline y.  planes (LDS): 4 x width bytes.  Wave c codes component c into
// stage + (4 y + c) * stage_pitch and stores the coded length in lengths[4 y + c].
__global__ __launch_bounds__(kBlock) void hdr_rows_kernel(const float* __restrict__ rgb, int width,
                                                          uint8_t* __restrict__ stage, uint64_t stage_pitch,
                                                          uint32_t* __restrict__ lengths) {
    extern __shared__ uint8_t planes[];
    const uint32_t y = blockIdx.x;
    const float* row = rgb + (size_t)y * width * 3;
    for (int x = threadIdx.x; x < width; x += kBlock) {
        const uint32_t v = rgbe_of(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
        planes[x] = (uint8_t)v;
        planes[width + x] = (uint8_t)(v >> 8);
        planes[2 * width + x] = (uint8_t)(v >> 16);
        planes[3 * width + x] = (uint8_t)(v >> 24);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t c = threadIdx.x >> 6;
    const uint8_t* b = planes + (size_t)c * width;
    uint8_t* o = stage + (size_t)(4 * y + c) * stage_pitch;
    const uint64_t le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);  // lanes <= this one
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t total = 0;      // bytes coded so far
    int run_start = 0;       // where the run of equal bytes that reaches into this block began
    int stretch_start = 0;   // where the current stretch of literal bytes began
    // Lane masks of the previous block (wave-uniform): eq bit l = byte l equals byte l + 1, tri bit l = bytes l, l + 1,
    // l + 2 exist and are equal, run bit l = byte l lies in a run of >= 3.
    uint64_t eq_prev = 0, tri_prev = 0, run_prev = 0;
    uint32_t cur = lane < width ? b[lane] : 0u, next = 0;
    for (int base = 0; base < width; base += 64) {
        const int i = base + lane;
        next = (i + 64 < width) ? b[i + 64] : 0u;
        const bool valid = i < width;
        // the first three bytes of the next block decide the last lanes' flags
        const uint32_t n0 = __shfl(next, 0), n1 = __shfl(next, 1), n2 = __shfl(next, 2);
        const uint32_t down = __shfl_down(cur, 1);
        const uint32_t p1 = lane == 63 ? n0 : down;
        const bool same_as_next = i + 1 < width && cur == p1;
        const uint64_t eq = __ballot(same_as_next);
        const uint64_t eq_n0 = (base + 65 < width && n0 == n1) ? 1ull : 0ull;  // eq of the next block's lanes 0, 1
        const uint64_t eq_n1 = (base + 66 < width && n1 == n2) ? 1ull : 0ull;
        const uint64_t tri = eq & ((eq >> 1) | (eq_n0 << 63));
        const uint64_t tri_n0 = eq_n0 & eq_n1;
        const uint64_t run = tri | (tri << 1) | (tri_prev >> 63) | (tri << 2) | (tri_prev >> 62);
        const uint64_t run_n0 = tri_n0 | (tri >> 63) | ((tri >> 62) & 1ull);
        const uint64_t before_in_run = (run << 1) | (run_prev >> 63);  // bit l: byte l - 1 lies in a run
        const uint64_t after_in_run = (run >> 1) | (run_n0 << 63);     // bit l: byte l + 1 lies in a run
        const uint64_t same_as_before = (eq << 1) | (eq_prev >> 63);
        const uint64_t valid_mask = __ballot(valid);
        const uint64_t rh = run & ~same_as_before;  // first byte of a run
        const uint64_t sh = valid_mask & ~run & (before_in_run | (base == 0 ? 1ull : 0ull));  // first literal byte of a stretch
        const bool in_run = (run >> lane) & 1ull;
        // start of the run / stretch this byte belongs to: the last start at or before it
        const uint64_t rh_le = rh & le_mask, sh_le = sh & le_mask;
        const int my_run_start = rh_le ? base + 63 - __clzll((long long)rh_le) : run_start;
        const int my_stretch_start = sh_le ? base + 63 - __clzll((long long)sh_le) : stretch_start;
        if (rh) run_start = base + 63 - __clzll((long long)rh);
        if (sh) stretch_start = base + 63 - __clzll((long long)sh);
        // bytes this byte emits: a packet's first byte emits the length byte as well
        const int k = in_run ? i - my_run_start : i - my_stretch_start;
        const int in_packet = in_run ? k % 127 : k % 128;
        const uint32_t emits = !valid ? 0u : (in_run ? (in_packet == 0 ? 2u : 0u) : (in_packet == 0 ? 2u : 1u));
        const uint64_t e1 = __ballot(emits >= 1u), e2 = __ballot(emits == 2u);
        const uint32_t off = total + (uint32_t)__popcll(e1 & lt_mask) + (uint32_t)__popcll(e2 & lt_mask);
        total += (uint32_t)__popcll(e1) + (uint32_t)__popcll(e2);
        if (in_run) {
            const bool last = in_packet == 126 || !same_as_next;
            if (last) {  // bytes after the packet's first emit nothing, so they all see the offset behind it
                const uint32_t at = in_packet == 0 ? off : off - 2u;
                o[at] = (uint8_t)(128 + in_packet + 1);
                o[at + 1] = (uint8_t)cur;
            }
        } else if (valid) {
            o[in_packet == 0 ? off + 1u : off] = (uint8_t)cur;
            const bool last = in_packet == 127 || i + 1 >= width || ((after_in_run >> lane) & 1ull);
            if (last) o[in_packet == 0 ? off : off - (uint32_t)in_packet - 1u] = (uint8_t)(in_packet + 1);
        }
        eq_prev = eq;
        tri_prev = tri;
        run_prev = run;
        cur = next;
    }
    if (lane == 0) lengths[4 * y + c] = total;
}

// Exclusive scan of the 4 * height coded lengths (one workgroup, 1024 entries per round) into file offsets: component
// s = 4 y + c starts at head.len + 4 (y + 1) + sum of the lengths before it.  Also writes the header and the size
// (flat_bytes: pixel bytes of the layout without run-length coding, which has no components to scan).
__global__ __launch_bounds__(1024) void hdr_offsets_kernel(const uint32_t* __restrict__ lengths, uint32_t n,
                                                           HdrHead head, uint64_t flat_bytes,
                                                           uint64_t* __restrict__ offsets, uint8_t* __restrict__ out,
                                                           uint64_t* __restrict__ size_out) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint64_t carry;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    if (t < head.len) out[t] = head.text[t];
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t s = base + t;
        const uint32_t len = s < n ? lengths[s] : 0u;
        uint32_t incl = len;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63) wave_sums[wave] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wave; w++) before += wave_sums[w];
        if (s < n) offsets[s] = (uint64_t)head.len + 4ull * (s / 4u + 1u) + before + (incl - len);
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
    if (t == 0) *size_out = (uint64_t)head.len + flat_bytes + 4ull * (n / 4u) + carry;
}

// One workgroup per component stream: staging slot -> file position; component 0 also writes the scanline marker.
__global__ __launch_bounds__(kBlock) void hdr_gather_kernel(const uint8_t* __restrict__ stage, uint64_t stage_pitch,
                                                            const uint32_t* __restrict__ lengths,
                                                            const uint64_t* __restrict__ offsets, int width,
                                                            uint8_t* __restrict__ out) {
    const uint32_t s = blockIdx.x;
    const uint8_t* src = stage + (size_t)s * stage_pitch;
    uint8_t* dst = out + offsets[s];
    const uint32_t len = lengths[s];
    if ((s & 3u) == 0 && threadIdx.x < 4) {
        const uint8_t marker[4] = {2, 2, (uint8_t)((width & 0xff00) >> 8), (uint8_t)(width & 0xff)};
        dst[(int)threadIdx.x - 4] = marker[threadIdx.x];
    }
    for (uint32_t i = threadIdx.x; i < len; i += kBlock) dst[i] = src[i];
}

}  // namespace

uint64_t hdr_bound(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    const uint64_t w = (uint64_t)width, h = (uint64_t)height;
    if (width < 8 || width >= 32768) return 128 + 4 * w * h;
    return 128 + h * (4 + 4 * (w + (w + 127) / 128));
}

int hdr_encode_device(rthip_ctx* ctx, const float* d_rgb, int width, int height, uint8_t* d_out, uint64_t capacity,
                      uint64_t* d_size, hipStream_t stream) {
    if (width <= 0 || height <= 0) return rthip_fail(ctx, RTHIP_E_INVALID, "image %d x %d", width, height);
    if (capacity < hdr_bound(width, height))
        return rthip_fail(ctx, RTHIP_E_LIMIT, "output buffer of %llu bytes is below rthip_hdr_bound = %llu",
                          (unsigned long long)capacity, (unsigned long long)hdr_bound(width, height));
    HdrHead head;
    const int hl = snprintf((char*)head.text, sizeof head.text,
                            "#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\n"
                            "EXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n", height, width);
    if (hl <= 0 || hl >= (int)sizeof head.text) return rthip_fail(ctx, RTHIP_E_INVALID, "header does not fit");
    head.len = (uint32_t)hl;
    const uint32_t n_streams = 4u * (uint32_t)height;
    RT_TRY(rthip_reserve(ctx, ctx->hdr_lengths, (size_t)n_streams * 4 + 16));
    RT_TRY(rthip_reserve(ctx, ctx->hdr_offsets, (size_t)n_streams * 8 + 16));
    if (width < 8 || width >= 32768) {  // no run-length coding: header, pixels
        const uint64_t n_pixels = (uint64_t)width * height;
        hdr_offsets_kernel<<<1, 1024, 0, stream>>>((const uint32_t*)ctx->hdr_lengths.p, 0u, head, 4 * n_pixels,
                                                   (uint64_t*)ctx->hdr_offsets.p, d_out, d_size);
        hdr_flat_kernel<<<(uint32_t)((n_pixels + kBlock - 1) / kBlock), kBlock, 0, stream>>>(d_rgb, n_pixels,
                                                                                              d_out + head.len);
        RT_HIP(ctx, hipGetLastError());
        return RTHIP_OK;
    }
    const uint64_t stage_pitch = ((uint64_t)width + ((uint64_t)width + 127) / 128 + 15) & ~15ull;
    RT_TRY(rthip_reserve(ctx, ctx->hdr_stage, (size_t)n_streams * stage_pitch));
    const size_t lds = 4 * (size_t)width;
    if (lds > 64 * 1024)
        RT_HIP(ctx, hipFuncSetAttribute((const void*)hdr_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hdr_rows_kernel<<<(uint32_t)height, kBlock, lds, stream>>>(d_rgb, width, (uint8_t*)ctx->hdr_stage.p, stage_pitch,
                                                              (uint32_t*)ctx->hdr_lengths.p);
    hdr_offsets_kernel<<<1, 1024, 0, stream>>>((const uint32_t*)ctx->hdr_lengths.p, n_streams, head, 0,
                                               (uint64_t*)ctx->hdr_offsets.p, d_out, d_size);
    hdr_gather_kernel<<<n_streams, kBlock, 0, stream>>>((const uint8_t*)ctx->hdr_stage.p, stage_pitch,
                                                        (const uint32_t*)ctx->hdr_lengths.p,
                                                        (const uint64_t*)ctx->hdr_offsets.p, width, d_out);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}
