// TEST INFRASTRUCTURE -- not part of the product.
//
// The reference's .hdr writer (stbi_write_hdr, src/third_party/stb/stb_image_write.h:794, which every demo calls
// with 3 components: src/demos/cornell/main.c:222) compiled unmodified into a shared library, writing into memory
// through the writer's own callback interface (stbi_write_hdr_to_func, stb_image_write.h:787) instead of a file.
// Contains no reference code; the header is #included from /root/reference/src (see oracle/Makefile).
// Output: oracle/_ref/libref_hdr.so.

#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "third_party/stb/stb_image_write.h"

#include <stdint.h>
#include <string.h>

struct RefSink {
    unsigned char* out;
    uint64_t cap, len;
};

static void ref_sink(void* context, void* data, int size) {
    RefSink* s = (RefSink*)context;
    if (s->len + (uint64_t)size <= s->cap) memcpy(s->out + s->len, data, (size_t)size);
    s->len += (uint64_t)size;
}

// Returns the number of bytes the writer produced (more than `cap`: the buffer was too small), 0 on failure.
extern "C" uint64_t ref_hdr_encode(const float* rgb, int width, int height, unsigned char* out, uint64_t cap) {
    RefSink s = {out, cap, 0};
    if (!stbi_write_hdr_to_func(ref_sink, &s, width, height, 3, rgb)) return 0;
    return s.len;
}
