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
// running positions (start of the current run, start of the current stretch), so a wave codes a component 256 bytes
// at a time, four consecutive bytes per lane: "equal to the next byte" of a lane's bytes and of its neighbours' tells
// every byte whether it lies in a run; ballots and shuffles give the two start positions and the output offset of every
// byte (a byte emits 0, 1 or 2 bytes), and the last byte of each packet writes the packet's length.
//
// Three launches: hdr_rows_kernel (one workgroup per scanline: RGBE conversion into LDS planes, then one wave per
// component codes its plane, 256 bytes a step, into a staging slot and records the length), hdr_offsets_kernel (scan of the lengths,
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

// One workgroup per scanline y.  planes (LDS): 4 planes of `pitch` bytes (room for a component or its coding, a multiple of
// four so that a lane reads its four bytes as one word).  Wave c codes component c into stage + (4 y + c) * stage_pitch and stores
// the coded length in lengths[4 y + c].
//
// A wave codes 256 bytes per step, lane l the four consecutive bytes 4 l .. 4 l + 3 of the block.  What a byte needs to
// know of its neighbourhood -- equal to the next byte (e), first of three equal bytes (tri), inside a run of >= 3 (run) --
// is four bits per lane, made from the lane's own word and the next lane's (one shuffle down; the last lane takes the
// first bytes of the next block) and handed to the next lane (two shuffles up; the first lane takes what the previous
// block's last lane left).  The start of the run / literal stretch a byte belongs to is the last head at or before it: the
// heads inside the lane are scanned in the lane, the one that reaches in from outside comes from the nearest earlier
// lane that has a head (one ballot and one shuffle per kind), or from the previous blocks.  A lane emits 0 to 8 bytes;
// their offsets are prefix sums over four ballots (one per bit of the lane's count).
__global__ __launch_bounds__(kBlock) void hdr_rows_kernel(const float* __restrict__ rgb, int width, int pitch,
                                                          uint8_t* __restrict__ stage, uint64_t stage_pitch,
                                                          uint32_t* __restrict__ lengths) {
    extern __shared__ uint8_t planes[];
    const uint32_t y = blockIdx.x;
    const float* row = rgb + (size_t)y * width * 3;
    for (int x = threadIdx.x; x < width; x += kBlock) {
        const uint32_t v = rgbe_of(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
        planes[x] = (uint8_t)v;
        planes[pitch + x] = (uint8_t)(v >> 8);
        planes[2 * pitch + x] = (uint8_t)(v >> 16);
        planes[3 * pitch + x] = (uint8_t)(v >> 24);
    }
    __syncthreads();
#ifdef RT_HDR_PHASE1_ONLY  // dev build: how long the conversion alone takes
    if (threadIdx.x < 4) lengths[4 * y + threadIdx.x] = planes[threadIdx.x * pitch];
    return;
#endif

    const int lane = threadIdx.x & 63;
    const uint32_t c = threadIdx.x >> 6;
    const uint32_t* b = (const uint32_t*)(planes + (size_t)c * pitch);  // word k holds bytes 4 k .. 4 k + 3
    // The coded bytes go back into the component's own plane (scattered byte stores cost far less in LDS than in global
    // memory) and from there to the staging slot as whole words.  Coding byte i writes no further than byte
    // i + i / 128 + 2 of the plane, and a step holds its own 256 bytes and the next step's in registers before it
    // stores, so nothing is overwritten before it has been read; the pitch leaves room for the growth.
    uint8_t* o = planes + (size_t)c * pitch;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t total = 0;      // bytes coded so far
    int run_start = 0;       // where the last run of equal bytes that began in an earlier block began
    int stretch_start = 0;   // where the last stretch of literal bytes that began in an earlier block began
    uint32_t carry = 0;      // the previous block's last lane: e | tri << 4 | run << 8 of its four bytes
    uint32_t prev_byte = 0;  // the previous block's last byte
    uint32_t w = 4 * lane < width ? b[lane] : 0u, w_next = 0;
    for (int base = 0; base < width; base += 256) {
        const int i0 = base + 4 * lane;
        w_next = (i0 + 256 < width) ? b[(i0 + 256) >> 2] : 0u;
        // A whole block of bytes that all equal the byte in front of them -- the inside of a long run: exponent planes,
        // black or saturated regions, flat backgrounds -- is recognised from the words alone and costs a few packet ends:
        // bytes base - 1, base, base + 1 are equal, so the run began at or before base - 1 and run_start knows it.
        if (base != 0 && base + 256 <= width) {
            uint32_t before = __shfl_up(w >> 24, 1);
            if (lane == 0) before = prev_byte;
            if (__ballot((w ^ ((w << 8) | before)) != 0u) == 0ull) {
                const uint32_t wn = (uint32_t)__builtin_amdgcn_readfirstlane((int)w_next);
                const uint32_t value = prev_byte;
                const bool e3 = base + 256 < width && (wn & 0xffu) == value;                               // the run goes on into the next block
                const bool e4 = base + 257 < width && (wn & 0xffu) == ((wn >> 8) & 0xffu);
                const uint32_t k0 = (uint32_t)(base - __builtin_amdgcn_readfirstlane(run_start));          // the block's first byte in its run
                const uint32_t q = k0 - ((k0 * 33027u) >> 22) * 127u;                                      // k0 mod 127, k0 < 32768: its place in its packet
                // the packet under way (or, with q == 0, the one that begins here) has its two bytes at `slot`
                uint32_t slot = q == 0u ? total : total - 2u;
                uint32_t next_total = q == 0u ? total + 2u : total;
                uint32_t end = 126u - q;  // offset in the block of that packet's 127th byte
                for (;;) {
                    if (end <= 255u) {  // a full packet ends inside the block
                        if (lane == 0) {
                            o[slot] = (uint8_t)255u;
                            o[slot + 1u] = (uint8_t)value;
                        }
                        if (end == 255u) break;
                        slot = next_total;  // the next packet begins behind it
                        next_total += 2u;
                        end += 127u;
                    } else {  // the packet reaches the block's end: it ends there if the run does
                        if (!e3 && lane == 0) {
                            uint32_t pl = q + 1u;  // (q + 255) mod 127 = (q + 1) mod 127: the last byte's place in its packet
                            pl = pl >= 127u ? pl - 127u : pl;
                            o[slot] = (uint8_t)(129u + pl);
                            o[slot + 1u] = (uint8_t)value;
                        }
                        break;
                    }
                }
                total = next_total;
                // what the next block's first lane needs to know of this block's last four bytes: e, tri, run
                carry = (7u | (e3 ? 8u : 0u)) | ((3u | (e3 ? 4u : 0u) | ((e3 && e4) ? 8u : 0u)) << 4) | (15u << 8);
                w = w_next;
                continue;
            }
        }
        // bytes 0..3 are the lane's own, 4..6 the next lane's first three (the next block's for the last lane)
        const uint32_t from_next = __shfl_down(w, 1);
        const uint32_t first_of_next_block = __shfl(w_next, 0);
        const uint32_t nx = lane == 63 ? first_of_next_block : from_next;
        uint32_t by[7];
#pragma unroll
        for (int j = 0; j < 4; j++) by[j] = (w >> (8 * j)) & 0xffu;
#pragma unroll
        for (int j = 0; j < 3; j++) by[4 + j] = (nx >> (8 * j)) & 0xffu;
        uint32_t e = 0;  // bit j: bytes i0 + j and i0 + j + 1 both exist and are equal (j = 0 .. 5)
#pragma unroll
        for (int j = 0; j < 6; j++) e |= (i0 + j + 1 < width && by[j] == by[j + 1]) ? (1u << j) : 0u;
        const uint32_t tri = e & (e >> 1);  // bit j: bytes j, j + 1, j + 2 are equal (j = 0 .. 4)
        // what the previous lane knows of its own bytes: e and tri now, run once this lane's run bits are made
        const uint32_t mine1 = (e & 15u) | ((tri & 15u) << 4);
        uint32_t prev1 = __shfl_up(mine1, 1);
        if (lane == 0) prev1 = carry & 0xffu;
        const uint32_t e_before = (prev1 >> 3) & 1u;                   // byte i0 - 1 equals byte i0
        const uint32_t tri_m1 = (prev1 >> 7) & 1u, tri_m2 = (prev1 >> 6) & 1u;  // tri of bytes i0 - 1 and i0 - 2
        // run bit j: byte j lies in a run of >= 3 (it is the first, second or third of three equal bytes), j = 0 .. 4
        const uint32_t tri_ext = (tri << 2) | (tri_m1 << 1) | tri_m2;  // bit j + 2: tri of byte j
        const uint32_t run = (tri_ext >> 2 | tri_ext >> 1 | tri_ext) & 31u;
        uint32_t prev_run = __shfl_up(run, 1);
        if (lane == 0) prev_run = carry >> 8;
        const uint32_t run_before = (prev_run >> 3) & 1u;  // byte i0 - 1 lies in a run
        uint32_t valid = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) valid |= (i0 + j < width) ? (1u << j) : 0u;
        const uint32_t same_as_before = ((e << 1) | e_before) & 15u;
        const uint32_t in_run_before = ((run << 1) | run_before) & 15u;
        const uint32_t rh_bits = run & ~same_as_before & 15u;
        // Two kinds of block make up most of an image and need far less than the general case below (same bytes out, by
        // the same rules): a block without a run in or next to it -- every byte a literal of the stretch that is under
        // way -- and a block that lies wholly inside one run that began earlier.  Both tests are wave-uniform.
        if (__ballot(run != 0u || run_before != 0u) == 0ull) {
            const uint32_t p0 = (uint32_t)(i0 - stretch_start) & 127u;  // place of the lane's first byte in its packet
            const uint32_t head_j = (128u - p0) & 127u;                  // which of the lane's bytes opens a packet (>= 4: none)
            const bool has_head = head_j < 4u && ((valid >> head_j) & 1u);
            const uint64_t heads = __ballot(has_head);
            uint32_t off = total + 4u * (uint32_t)lane + (uint32_t)__popcll(heads & lt_mask);  // (lanes in front of a valid byte are full)
            const uint32_t in_block = (uint32_t)(width - base < 256 ? width - base : 256);
            total += in_block + (uint32_t)__popcll(heads);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool ok = (valid >> j) & 1u;
                const uint32_t p = (p0 + (uint32_t)j) & 127u;
                const bool first = p == 0u;
                const bool last = p == 127u || i0 + j + 1 >= width;
                if (ok) o[first ? off + 1u : off] = (uint8_t)by[j];
                if (ok && last) o[first ? off : off - p - 1u] = (uint8_t)(p + 1u);
                off += ok ? (first ? 2u : 1u) : 0u;
            }
            carry = (uint32_t)__shfl((int)mine1, 63);  // (its run bits are zero)
            prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)w, 63) >> 24;
            w = w_next;
            continue;
        }
        if (__ballot((valid & ~run) != 0u || rh_bits != 0u) == 0ull) {
            // every byte of the block continues the run that began at run_start: only packet ends store anything
            const uint32_t k0 = (uint32_t)(i0 - run_start);
            const uint32_t p0 = k0 - __umul24(__umul24(k0, 33027u) >> 22, 127u);  // k0 mod 127 (see below)
            const uint32_t head_j = (127u - p0) % 127u;  // which byte of the lane opens a packet (>= 4: none); p0 < 127
            const bool has_head = head_j < 4u && ((valid >> head_j) & 1u);
            const uint64_t heads = __ballot(has_head);
            uint32_t off = total + 2u * (uint32_t)__popcll(heads & lt_mask);
            total += 2u * (uint32_t)__popcll(heads);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool ok = (valid >> j) & 1u;
                uint32_t p = p0 + (uint32_t)j;
                p = p >= 127u ? p - 127u : p;
                const bool first = p == 0u;
                const bool last = p == 126u || !((e >> j) & 1u);
                const uint32_t at = first ? off : off - 2u;
                if (ok && last) {
                    o[at] = (uint8_t)(129u + p);
                    o[at + 1u] = (uint8_t)by[j];
                }
                off += (ok && first) ? 2u : 0u;
            }
            carry = (uint32_t)__shfl((int)(mine1 | ((run & 15u) << 8)), 63);
            prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)w, 63) >> 24;
            w = w_next;
            continue;
        }
        const uint32_t rh = rh_bits;  // first byte of a run
        const uint32_t sh = valid & ~run & (in_run_before | (i0 == 0 ? 1u : 0u)) & 15u;  // first literal byte of a stretch
        // the heads that reach in from outside the lane: the last one in an earlier lane of this block, else the carried one
        const uint64_t rh_lanes = __ballot(rh != 0u), sh_lanes = __ballot(sh != 0u);
        const int my_last_rh = i0 + 31 - __clz(rh ? (int)rh : 1);  // the lane's last head (only read from lanes that have one)
        const int my_last_sh = i0 + 31 - __clz(sh ? (int)sh : 1);
        const uint64_t rh_before = rh_lanes & lt_mask, sh_before = sh_lanes & lt_mask;
        const int rh_src = rh_before ? 63 - __clzll((long long)rh_before) : 0;
        const int sh_src = sh_before ? 63 - __clzll((long long)sh_before) : 0;
        const int rh_in = __shfl(my_last_rh, rh_src), sh_in = __shfl(my_last_sh, sh_src);
        int rs = rh_before ? rh_in : run_start, ss = sh_before ? sh_in : stretch_start;
        if (rh_lanes) run_start = __shfl(my_last_rh, 63 - __clzll((long long)rh_lanes));
        if (sh_lanes) stretch_start = __shfl(my_last_sh, 63 - __clzll((long long)sh_lanes));
        // per byte: place in its packet, bytes emitted
        int in_packet[4];
        uint32_t emits[4], lane_sum = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((rh >> j) & 1u) rs = i0 + j;
            if ((sh >> j) & 1u) ss = i0 + j;
            const bool in_run = (run >> j) & 1u;
            const uint32_t k = (uint32_t)(i0 + j - (in_run ? rs : ss));  // < 32768: rows of that width are not run-length coded
            // k mod 127 without a division, both forms computed and one selected (no branch): floor(k / 127) =
            // k * 33027 >> 22 for k < 33554 (33027 = ceil(2^22 / 127)), with 24-bit multiplies
            const uint32_t mod127 = k - __umul24(__umul24(k, 33027u) >> 22, 127u);
            const uint32_t p = in_run ? mod127 : (k & 127u);
            in_packet[j] = (int)p;
            const uint32_t ok = (valid >> j) & 1u;
            emits[j] = ok * ((p == 0u ? 2u : 0u) | ((p != 0u && !in_run) ? 1u : 0u));
            lane_sum += emits[j];
        }
        uint32_t off = total;
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            const uint64_t m = __ballot((lane_sum >> bit) & 1u);
            off += (uint32_t)__popcll(m & lt_mask) << bit;
            total += (uint32_t)__popcll(m) << bit;
        }
        // Every byte stores at most twice -- its own value, and, where it ends a packet, the packet's length byte -- at
        // places and under conditions computed without branches.  In a run only the packet's last byte stores (count at
        // `at`, value behind it): the bytes after a packet's first emit nothing, so they all see the offset behind it.
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool ok = (valid >> j) & 1u;
            const int i = i0 + j;
            const bool in_run = (run >> j) & 1u;
            const bool same_as_next = (e >> j) & 1u;
            const uint32_t p = (uint32_t)in_packet[j];
            const bool first = p == 0u;
            const bool last_run = p == 126u || !same_as_next;
            const bool last_lit = p == 127u || i + 1 >= width || ((run >> (j + 1)) & 1u);
            const uint32_t at = first ? off : off - 2u;                 // a run packet's first byte
            const uint32_t lit_at = first ? off + 1u : off;             // a literal byte's place
            const uint32_t lit_head = first ? off : off - p - 1u;       // its packet's length byte
            const bool value_en = ok && (in_run ? last_run : true);
            const bool count_en = ok && (in_run ? last_run : last_lit);
            const uint32_t value_at = in_run ? at + 1u : lit_at, count_at = in_run ? at : lit_head;
            const uint32_t count = in_run ? 129u + p : p + 1u;
            if (value_en) o[value_at] = (uint8_t)by[j];
            if (count_en) o[count_at] = (uint8_t)count;
            off += emits[j];
        }
        carry = __shfl((int)(mine1 | ((run & 15u) << 8)), 63);
        prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)w, 63) >> 24;
        w = w_next;
    }
    if (lane == 0) lengths[4 * y + c] = total;
    uint32_t* slot = (uint32_t*)(stage + (size_t)(4 * y + c) * stage_pitch);
    for (uint32_t k = lane; 4u * k < total; k += 64u) slot[k] = b[k];  // (the slot and the plane both end on a whole word)
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
// The file position is any byte address, the staging slot is 16-byte aligned: the bytes up to the first aligned word of
// the destination and behind the last one go one by one, the words between as words, each put together from the two
// aligned source words it straddles.
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
    const uint32_t head = (4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u;  // bytes in front of the first aligned word
    const uint32_t lead = head < len ? head : len;
    const uint32_t words = (len - lead) / 4u;
    if (threadIdx.x < lead) dst[threadIdx.x] = src[threadIdx.x];
    const uint32_t* src_w = (const uint32_t*)src;
    uint32_t* dst_w = (uint32_t*)(dst + lead);
    const uint32_t shift = 8u * (lead & 3u);  // source byte lead + 4 k lies `lead` bytes into source word k + lead / 4
    for (uint32_t k = threadIdx.x; k < words; k += kBlock) {
        const uint32_t lo = src_w[k + lead / 4u];
        const uint32_t hi = shift ? src_w[k + lead / 4u + 1u] : 0u;  // (within the slot: its pitch leaves room behind every coded length)
        dst_w[k] = shift ? (lo >> shift) | (hi << (32u - shift)) : lo;
    }
    const uint32_t done = lead + 4u * words;
    if (threadIdx.x < len - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
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
    const uint64_t stage_pitch = ((uint64_t)width + ((uint64_t)width + 127) / 128 + 31) & ~15ull;  // the longest coding, a word of slack behind it (hdr_gather_kernel reads whole words), 16-byte aligned
    RT_TRY(rthip_reserve(ctx, ctx->hdr_stage, (size_t)n_streams * stage_pitch));
    const int pitch = (width + (width + 127) / 128 + 2 + 3) & ~3;  // a component, or its coding (hdr_rows_kernel codes in place)
    const size_t lds = 4 * (size_t)pitch;
    if (lds > 64 * 1024)
        RT_HIP(ctx, hipFuncSetAttribute((const void*)hdr_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hdr_rows_kernel<<<(uint32_t)height, kBlock, lds, stream>>>(d_rgb, width, pitch, (uint8_t*)ctx->hdr_stage.p, stage_pitch,
                                                              (uint32_t*)ctx->hdr_lengths.p);
    hdr_offsets_kernel<<<1, 1024, 0, stream>>>((const uint32_t*)ctx->hdr_lengths.p, n_streams, head, 0,
                                               (uint64_t*)ctx->hdr_offsets.p, d_out, d_size);
    hdr_gather_kernel<<<n_streams, kBlock, 0, stream>>>((const uint8_t*)ctx->hdr_stage.p, stage_pitch,
                                                        (const uint32_t*)ctx->hdr_lengths.p,
                                                        (const uint64_t*)ctx->hdr_offsets.p, width, d_out);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}
