// Wavefront OBJ -> indexed triangle mesh, the result of the reference's ms_load_obj (src/mesh/mesh.c:73-219): vertices
// are the distinct (position, normal, uv) index triples of the faces, numbered in the order in which they first appear
// (ms_add_to_vertex_map, mesh.c:19-37), one 16-byte slot per attribute (geo_vertex_i_offset, src/geo/geo.c:2-18).
//
// The reference reads the file twice line by line with sscanf and deduplicates through a chained hash map, all on one
// core.  Here the text is split at line boundaries over the host's cores (count pass, prefix sums, parse pass: the
// same two passes, so every thread knows where its records go), and the deduplication runs on the GPU:
//   key(corner) = (p, n, t) packed into 64 bits, value = corner number;  stable radix sort (the LBVH builder's);
//   the first element of every run of equal keys is the corner where that triple first appears;
//   a prefix sum over "corner is a first appearance" numbers the vertices in first-appearance order;
//   every corner takes the number of its run's first corner;  first corners gather their attributes.
// The vertex data is produced as SoA (what the BLAS builder reads) and, for callers that want the reference's
// MS_Mesh, in its interleaved layout.

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <charconv>
#include <chrono>
#include <thread>

#include "rthip_internal.h"

namespace {

constexpr int kBlock = 256;
inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// ---- host: text -> arrays ---------------------------------------------------------------------------------------

struct Chunk {
    size_t begin = 0, end = 0;              // byte range, begins at a line start
    uint64_t nv = 0, nvn = 0, nvt = 0, nf = 0;  // records in the range
    uint64_t ov = 0, ovn = 0, ovt = 0, of = 0;  // records before the range
    uint64_t first_line = 0, lines = 0;
    uint64_t bad_line = 0;                  // 1-based line of the first face that does not parse (0: none)
    uint32_t max_p = 0, max_n = 0, max_t = 0;
};

inline bool begins(const char* s, const char* e, const char* prefix, size_t len) {
    return (size_t)(e - s) >= len && memcmp(s, prefix, len) == 0;
}

void count_chunk(const char* text, Chunk& c) {
    const char* s = text + c.begin;
    const char* end = text + c.end;
    while (s < end) {
        const char* nl = (const char*)memchr(s, '\n', (size_t)(end - s));
        const char* e = nl ? nl : end;
        if (begins(s, e, "v ", 2)) c.nv++;
        else if (begins(s, e, "vn ", 3)) c.nvn++;
        else if (begins(s, e, "vt ", 3)) c.nvt++;
        else if (begins(s, e, "f ", 2)) c.nf++;
        c.lines++;
        s = nl ? nl + 1 : end;
    }
}

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

// "%f" of sscanf = strtof.  std::from_chars is correctly rounded like strtof and several times faster, but differs at
// the edges (no leading '+', no hexadecimal floats, out-of-range values reported instead of rounded), so anything
// unusual goes through strtof itself on a NUL-terminated copy.
inline bool scan_float(const char*& s, const char* e, float& out) {
    while (s < e && is_blank(*s)) s++;
    const char* t = s;
    if (t < e && *t == '+') t++;
    if (t < e && !(t != s && (*t == '+' || *t == '-'))) {
        float v = 0.f;
        const auto r = std::from_chars(t, e, v);
        if (r.ec == std::errc() && !(r.ptr < e && (*r.ptr == 'x' || *r.ptr == 'X'))) {
            out = v;
            s = r.ptr;
            return true;
        }
    }
    char buf[256];
    const size_t len = (size_t)(e - s) < sizeof buf - 1 ? (size_t)(e - s) : sizeof buf - 1;
    memcpy(buf, s, len);
    buf[len] = 0;
    char* after = nullptr;
    const float v = strtof(buf, &after);
    if (after == buf) return false;
    out = v;
    s += after - buf;
    return true;
}

inline void parse_floats(const char* s, const char* e, float* out, int n) {
    for (int k = 0; k < n; k++)
        if (!scan_float(s, e, out[k])) return;  // sscanf stops at the first field that does not convert; the rest stays 0
}

// " %u" of sscanf = strtoul: plain digit strings directly, signs and overflow through strtoul
inline bool scan_uint(const char*& s, const char* e, uint32_t& out) {
    while (s < e && is_blank(*s)) s++;
    const char* t = s;
    uint64_t v = 0;
    while (t < e && *t >= '0' && *t <= '9' && t - s < 10) v = v * 10 + (uint64_t)(*t++ - '0');
    if (t > s && !(t < e && *t >= '0' && *t <= '9') && v <= 0xffffffffull) {
        out = (uint32_t)v;
        s = t;
        return true;
    }
    if (!(s < e && ((*s >= '0' && *s <= '9') || *s == '+' || *s == '-'))) return false;
    char buf[64];
    const size_t len = (size_t)(e - s) < sizeof buf - 1 ? (size_t)(e - s) : sizeof buf - 1;
    memcpy(buf, s, len);
    buf[len] = 0;
    char* after = nullptr;
    const unsigned long v2 = strtoul(buf, &after, 10);
    if (after == buf) return false;
    out = (uint32_t)v2;
    s += after - buf;
    return true;
}

inline bool at(const char* s, const char* e, char c) { return s < e && *s == c; }

void parse_chunk(const char* text, Chunk& c, uint32_t attrs, float* pos, float* nrm, float* uv, uint32_t* cp, uint32_t* cn,
                 uint32_t* ct) {
    const char* s = text + c.begin;
    const char* end = text + c.end;
    uint64_t iv = c.ov, ivn = c.ovn, ivt = c.ovt, ifc = c.of, line_no = c.first_line;
    while (s < end) {
        const char* nl = (const char*)memchr(s, '\n', (size_t)(end - s));
        const char* e = nl ? nl : end;
        line_no++;
        if (begins(s, e, "v ", 2)) parse_floats(s + 1, e, pos + 3 * iv++, 3);
        else if (begins(s, e, "vn ", 3)) parse_floats(s + 2, e, nrm + 3 * ivn++, 3);
        else if (begins(s, e, "vt ", 3)) parse_floats(s + 2, e, uv + 2 * ivt++, 2);
        else if (begins(s, e, "f ", 2)) {
            const char* q = s + 1;
            bool ok = true;
            for (int k = 0; k < 3 && ok; k++) {  // mesh.c:145-181: the format follows the requested attributes
                uint32_t p = 0, n = 0, t = 0;
                ok = scan_uint(q, e, p);
                if (ok && attrs == 5u) ok = at(q, e, '/') && (++q, scan_uint(q, e, t));                             // " %u/%u"
                if (ok && attrs == 3u) ok = at(q, e, '/') && at(q + 1, e, '/') && (q += 2, scan_uint(q, e, n));     // " %u//%u"
                if (ok && attrs == 7u)                                                                              // " %u/%u/%u"
                    ok = at(q, e, '/') && (++q, scan_uint(q, e, t)) && at(q, e, '/') && (++q, scan_uint(q, e, n));
                cp[3 * ifc + k] = p;
                cn[3 * ifc + k] = n;
                ct[3 * ifc + k] = t;
                if (p > c.max_p) c.max_p = p;
                if (n > c.max_n) c.max_n = n;
                if (t > c.max_t) c.max_t = t;
            }
            if (!ok && c.bad_line == 0) c.bad_line = line_no;
            ifc++;
        }
        s = nl ? nl + 1 : end;
    }
}

template <typename F>
void run_threads(int n, F&& f) {
    std::vector<std::thread> th;
    for (int k = 1; k < n; k++) th.emplace_back([&f, k] { f(k); });
    f(0);
    for (auto& t : th) t.join();
}

inline int bits_for(uint64_t max_value) {
    int b = 0;
    while (b < 64 && (max_value >> b) != 0) b++;
    return b;
}

// ---- device: deduplication -----------------------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void obj_pack_kernel(const uint32_t* __restrict__ cp, const uint32_t* __restrict__ cn,
                                                          const uint32_t* __restrict__ ct, uint64_t m, int shift_p, int shift_n,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    keys[i] = ((uint64_t)cp[i] << shift_p) | ((uint64_t)cn[i] << shift_n) | (uint64_t)ct[i];
    vals[i] = (uint32_t)i;
}

// head[j] = 1 where sorted element j starts a run of equal keys; first[c] = 1 for the corner such an element came from
__global__ __launch_bounds__(kBlock) void obj_heads_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                           uint64_t m, uint32_t* __restrict__ head, uint32_t* __restrict__ first) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= m) return;
    const uint32_t h = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
    head[j] = h;
    first[vals[j]] = h;  // every corner is written exactly once
}

// run_first[r] = the corner that starts run r (head_scan is the inclusive prefix sum of head)
__global__ __launch_bounds__(kBlock) void obj_run_first_kernel(const uint32_t* __restrict__ head_scan,
                                                               const uint32_t* __restrict__ vals, uint64_t m,
                                                               uint32_t* __restrict__ run_first) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= m) return;
    const uint32_t r = head_scan[j];
    if (j == 0 || head_scan[j - 1] != r) run_first[r - 1] = vals[j];
}

// a corner's vertex = the number its run's first corner got (first_scan: inclusive prefix sum of first, by corner)
__global__ __launch_bounds__(kBlock) void obj_indices_kernel(const uint32_t* __restrict__ head_scan,
                                                             const uint32_t* __restrict__ vals,
                                                             const uint32_t* __restrict__ run_first,
                                                             const uint32_t* __restrict__ first_scan, uint64_t m,
                                                             uint32_t* __restrict__ indices) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= m) return;
    indices[vals[j]] = first_scan[run_first[head_scan[j] - 1]] - 1u;
}

// first corners gather their attributes (1-based indices, 0 = absent; mesh.c:39-70): SoA for the BLAS builder and the
// reference's interleaved layout (one 16-byte slot per attribute)
__global__ __launch_bounds__(kBlock) void obj_vertices_kernel(const uint32_t* __restrict__ first_scan, uint64_t m, uint32_t attrs,
                                                              const uint32_t* __restrict__ cp, const uint32_t* __restrict__ cn,
                                                              const uint32_t* __restrict__ ct, const float* __restrict__ pos,
                                                              const float* __restrict__ nrm, const float* __restrict__ uv,
                                                              float* __restrict__ px, float* __restrict__ py,
                                                              float* __restrict__ pz, float* __restrict__ nx,
                                                              float* __restrict__ ny, float* __restrict__ nz,
                                                              float* __restrict__ interleaved, uint32_t slots) {
    const uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= m) return;
    const uint32_t s = first_scan[c];
    if (!(c == 0 ? s != 0 : s != first_scan[c - 1])) return;
    const uint32_t v = s - 1u;
    float4* out = (float4*)interleaved + (size_t)v * slots;
    uint32_t slot = 0;
    if (attrs & 1u) {
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cp[c] != 0) {
            const float* q = pos + 3 * (size_t)(cp[c] - 1u);
            p = make_float4(q[0], q[1], q[2], 0.f);
        }
        px[v] = p.x;
        py[v] = p.y;
        pz[v] = p.z;
        out[slot++] = p;
    }
    if (attrs & 2u) {
        float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cn[c] != 0) {
            const float* q = nrm + 3 * (size_t)(cn[c] - 1u);
            n = make_float4(q[0], q[1], q[2], 0.f);
        }
        nx[v] = n.x;
        ny[v] = n.y;
        nz[v] = n.z;
        out[slot++] = n;
    }
    if (attrs & 4u) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ct[c] != 0) {
            const float* q = uv + 2 * (size_t)(ct[c] - 1u);
            t = make_float4(q[0], q[1], 0.f, 0.f);
        }
        out[slot++] = t;
    }
}

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

int obj_load(rthip_ctx* ctx, const char* path, uint32_t attrs, rthip_obj_info* info) {
    if (attrs == 0) attrs = 7u;  // mesh.c:85-87: position, normal and uv
    if (attrs != 1u && attrs != 3u && attrs != 5u && attrs != 7u)
        return rthip_fail(ctx, RTHIP_E_INVALID, "vertex attributes %u: positions are required (P = 1, N = 2, T = 4)", attrs);
    ObjState& o = ctx->obj;
    o.loaded = false;
    const auto t_start = std::chrono::steady_clock::now();
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return rthip_fail(ctx, RTHIP_E_INVALID, "Failed to open file %s", path);  // mesh.c:76-78
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return rthip_fail(ctx, RTHIP_E_INVALID, "cannot stat %s", path);
    }
    const size_t size = (size_t)st.st_size;
    const char* text = size ? (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    close(fd);
    if (text == (const char*)MAP_FAILED) return rthip_fail(ctx, RTHIP_E_NOMEM, "cannot map %s", path);

    // split at line starts, one range per thread
    int n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads > 16) n_threads = 16;  // one GPU's share of the host
    if (n_threads < 1) n_threads = 1;
    if (size < (1u << 20)) n_threads = 1;
    std::vector<Chunk> chunks((size_t)n_threads);
    size_t at = 0;
    for (int k = 0; k < n_threads; k++) {
        chunks[k].begin = at;
        size_t stop = k + 1 == n_threads ? size : size / n_threads * (size_t)(k + 1);
        if (stop < at) stop = at;
        if (stop < size) {
            const char* nl = (const char*)memchr(text + stop, '\n', size - stop);
            stop = nl ? (size_t)(nl - text) + 1 : size;
        }
        chunks[k].end = stop;
        at = stop;
    }
    // The reference reads with `while (!feof) { fgets; ... }` (mesh.c:108-109, 140-141): after the last line of a file
    // that ends in a newline, end-of-file has not been seen yet, the next fgets fails and leaves the buffer as it was,
    // and the loop body runs once more on the last line.  Such a file's last line therefore counts twice, in both
    // passes (a trailing face becomes a duplicated triangle): one more range that covers the last line again.
    if (size > 0 && text[size - 1] == '\n') {
        size_t last = size - 1;
        while (last > 0 && text[last - 1] != '\n') last--;
        Chunk again;
        again.begin = last;
        again.end = size;
        chunks.push_back(again);
    }
    const int n_ranges = (int)chunks.size();
    auto for_ranges = [&](auto&& f) {  // range k goes to thread k; the repeated last line to thread 0
        run_threads(n_threads, [&](int k) {
            f(chunks[k]);
            if (k == 0 && n_ranges > n_threads) f(chunks[n_threads]);
        });
    };
    for_ranges([&](Chunk& c) { count_chunk(text, c); });
    uint64_t nv = 0, nvn = 0, nvt = 0, nf = 0, lines = 0;
    for (auto& c : chunks) {
        c.ov = nv, c.ovn = nvn, c.ovt = nvt, c.of = nf, c.first_line = lines;
        nv += c.nv, nvn += c.nvn, nvt += c.nvt, nf += c.nf, lines += c.lines;
    }
    const uint64_t m = 3 * nf;  // corners
    if (m > (1ull << 31) || nv > (1ull << 31) || nvn > (1ull << 31) || nvt > (1ull << 31)) {
        if (size) munmap((void*)text, size);
        return rthip_fail(ctx, RTHIP_E_LIMIT, "%s: more than 2^31 records of one kind", path);
    }
    std::vector<float> pos(3 * nv + 1, 0.f), nrm(3 * nvn + 1, 0.f), uv(2 * nvt + 1, 0.f);
    std::vector<uint32_t> cp(m + 1, 0u), cn(m + 1, 0u), ct(m + 1, 0u);
    for_ranges([&](Chunk& c) { parse_chunk(text, c, attrs, pos.data(), nrm.data(), uv.data(), cp.data(), cn.data(), ct.data()); });
    if (size) munmap((void*)text, size);
    uint32_t max_p = 0, max_n = 0, max_t = 0;
    for (auto& c : chunks) {
        if (c.bad_line) return rthip_fail(ctx, RTHIP_E_INVALID, "%s:%llu: face does not match the requested attributes", path,
                                          (unsigned long long)c.bad_line);
        max_p = c.max_p > max_p ? c.max_p : max_p;
        max_n = c.max_n > max_n ? c.max_n : max_n;
        max_t = c.max_t > max_t ? c.max_t : max_t;
    }
    if (max_p > nv || max_n > nvn || max_t > nvt)
        return rthip_fail(ctx, RTHIP_E_INVALID, "%s: a face refers to a record the file does not hold", path);
    const double parse_ms = ms_since(t_start);

    // ---- GPU ----
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t slots = (attrs & 1u) + ((attrs >> 1) & 1u) + ((attrs >> 2) & 1u);
    o.attrs = attrs;
    o.index_count = (uint32_t)m;
    o.vertex_count = 0;
    float gpu_ms = 0.f;
    if (m != 0) {
        const int bt = bits_for(max_t), bn = bits_for(max_n), bp = bits_for(max_p);
        if (bt + bn + bp > 64)
            return rthip_fail(ctx, RTHIP_E_LIMIT, "%s: position, normal and uv indices need %d bits together, the key holds 64",
                              path, bt + bn + bp);
        LbvhWorkspace& w = ctx->ws;
        RT_TRY(rthip_reserve(ctx, w.keys[0], m * 8));
        RT_TRY(rthip_reserve(ctx, w.keys[1], m * 8));
        RT_TRY(rthip_reserve(ctx, w.vals[0], m * 4));
        RT_TRY(rthip_reserve(ctx, w.vals[1], m * 4));
        RT_TRY(rthip_reserve(ctx, o.cp, m * 4));
        RT_TRY(rthip_reserve(ctx, o.cn, m * 4));
        RT_TRY(rthip_reserve(ctx, o.ct, m * 4));
        RT_TRY(rthip_reserve(ctx, o.pos, (3 * nv + 1) * 4));
        RT_TRY(rthip_reserve(ctx, o.nrm, (3 * nvn + 1) * 4));
        RT_TRY(rthip_reserve(ctx, o.uv, (2 * nvt + 1) * 4));
        RT_TRY(rthip_reserve(ctx, o.head, m * 4));
        RT_TRY(rthip_reserve(ctx, o.first, m * 4));
        RT_TRY(rthip_reserve(ctx, o.run_first, m * 4));
        RT_TRY(rthip_reserve(ctx, o.indices, m * 4));
        RT_HIP(ctx, hipMemcpyAsync(o.cp.p, cp.data(), m * 4, hipMemcpyHostToDevice, s));
        RT_HIP(ctx, hipMemcpyAsync(o.cn.p, cn.data(), m * 4, hipMemcpyHostToDevice, s));
        RT_HIP(ctx, hipMemcpyAsync(o.ct.p, ct.data(), m * 4, hipMemcpyHostToDevice, s));
        RT_HIP(ctx, hipMemcpyAsync(o.pos.p, pos.data(), (3 * nv + 1) * 4, hipMemcpyHostToDevice, s));
        RT_HIP(ctx, hipMemcpyAsync(o.nrm.p, nrm.data(), (3 * nvn + 1) * 4, hipMemcpyHostToDevice, s));
        RT_HIP(ctx, hipMemcpyAsync(o.uv.p, uv.data(), (2 * nvt + 1) * 4, hipMemcpyHostToDevice, s));
        RT_HIP(ctx, hipEventRecord(ctx->ev0, s));
        const uint32_t blocks = div_up(m, kBlock);
        obj_pack_kernel<<<blocks, kBlock, 0, s>>>((const uint32_t*)o.cp.p, (const uint32_t*)o.cn.p, (const uint32_t*)o.ct.p, m,
                                                  bp ? bn + bt : 0, bn ? bt : 0, (uint64_t*)w.keys[0].p, (uint32_t*)w.vals[0].p);
        RT_HIP(ctx, hipGetLastError());
        int cur = 0;
        int n_bytes = (bt + bn + bp + 7) / 8;
        if (n_bytes < 1) n_bytes = 1;
        RT_TRY(radix_sort_workspace_device(ctx, m, n_bytes, s, &cur));
        const uint64_t* keys = (const uint64_t*)w.keys[cur].p;
        const uint32_t* vals = (const uint32_t*)w.vals[cur].p;
        obj_heads_kernel<<<blocks, kBlock, 0, s>>>(keys, vals, m, (uint32_t*)o.head.p, (uint32_t*)o.first.p);
        RT_HIP(ctx, hipGetLastError());
        RT_TRY(scan_inclusive_u32_device(ctx, (uint32_t*)o.head.p, m, s));
        RT_TRY(scan_inclusive_u32_device(ctx, (uint32_t*)o.first.p, m, s));
        obj_run_first_kernel<<<blocks, kBlock, 0, s>>>((const uint32_t*)o.head.p, vals, m, (uint32_t*)o.run_first.p);
        obj_indices_kernel<<<blocks, kBlock, 0, s>>>((const uint32_t*)o.head.p, vals, (const uint32_t*)o.run_first.p,
                                                     (const uint32_t*)o.first.p, m, (uint32_t*)o.indices.p);
        RT_HIP(ctx, hipGetLastError());
        uint32_t n_vertices = 0;
        RT_HIP(ctx, hipMemcpyAsync(&n_vertices, (const uint32_t*)o.first.p + (m - 1), 4, hipMemcpyDeviceToHost, s));
        RT_HIP(ctx, hipStreamSynchronize(s));
        o.vertex_count = n_vertices;
        for (DevBuf* b : {&o.px, &o.py, &o.pz, &o.nx, &o.ny, &o.nz}) RT_TRY(rthip_reserve(ctx, *b, (size_t)n_vertices * 4 + 16));
        RT_TRY(rthip_reserve(ctx, o.interleaved, (size_t)n_vertices * slots * 16 + 16));
        obj_vertices_kernel<<<blocks, kBlock, 0, s>>>((const uint32_t*)o.first.p, m, attrs, (const uint32_t*)o.cp.p,
                                                      (const uint32_t*)o.cn.p, (const uint32_t*)o.ct.p, (const float*)o.pos.p,
                                                      (const float*)o.nrm.p, (const float*)o.uv.p, (float*)o.px.p, (float*)o.py.p,
                                                      (float*)o.pz.p, (float*)o.nx.p, (float*)o.ny.p, (float*)o.nz.p,
                                                      (float*)o.interleaved.p, slots);
        RT_HIP(ctx, hipGetLastError());
        RT_HIP(ctx, hipEventRecord(ctx->ev1, s));
        RT_HIP(ctx, hipStreamSynchronize(s));
        RT_HIP(ctx, hipEventElapsedTime(&gpu_ms, ctx->ev0, ctx->ev1));
    }
    o.loaded = true;
    if (info) {
        info->vertex_count = o.vertex_count;
        info->index_count = o.index_count;
        info->attrs = attrs;
        info->vertex_bytes = slots * 16u;
        info->parse_ms = (float)parse_ms;
        info->dedup_ms = gpu_ms;
        info->total_ms = (float)ms_since(t_start);
        info->host_threads = (uint32_t)n_threads;
    }
    return RTHIP_OK;
}

int obj_read_back(rthip_ctx* ctx, float* out_vertices, uint32_t* out_indices) {
    ObjState& o = ctx->obj;
    if (!o.loaded) return rthip_fail(ctx, RTHIP_E_INVALID, "no OBJ file has been loaded");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t slots = (o.attrs & 1u) + ((o.attrs >> 1) & 1u) + ((o.attrs >> 2) & 1u);
    if (out_vertices && o.vertex_count)
        RT_HIP(ctx, hipMemcpy(out_vertices, o.interleaved.p, (size_t)o.vertex_count * slots * 16, hipMemcpyDeviceToHost));
    if (out_indices && o.index_count)
        RT_HIP(ctx, hipMemcpy(out_indices, o.indices.p, (size_t)o.index_count * 4, hipMemcpyDeviceToHost));
    return RTHIP_OK;
}
