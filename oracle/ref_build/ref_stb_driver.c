/* oracle/ref_build/ref_stb_driver.c - TEST INFRASTRUCTURE ONLY. Compiles the image decoder the reference vendors
 * (stb_image.h, used through tinygltf for every embedded texture) from where it lies under /root/reference, so that
 * tests can compare the product's PNG / JPEG decoders (csrc/srx_gltf.cpp) byte for byte with what the reference loads.
 * Nothing of the header is copied: it is included by path (-I$(REF)). */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO
#include "stb_image.h"
#include <string.h>

/* returns 0 and writes w, h, n (components in the file) and, when out is not NULL, w*h*4 bytes (4 components requested,
 * as tinygltf asks); 1 when stb_image refuses the data */
int ref_stb_decode(const unsigned char* data, int len, int* w, int* h, int* n, unsigned char* out) {
    unsigned char* px = stbi_load_from_memory(data, len, w, h, n, 4);
    if (!px) return 1;
    if (out) memcpy(out, px, (size_t)(*w) * (size_t)(*h) * 4);
    stbi_image_free(px);
    return 0;
}
