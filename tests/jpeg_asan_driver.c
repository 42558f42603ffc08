/* Test driver (tests/test_textures.py): feeds every case of a corpus file -- [u32 little-endian length][bytes] repeated --
 * to pyr_jpeg_decode. Built together with pyrite_amd/csrc/jpeg.c under -fsanitize=address,undefined: any out-of-bounds
 * access, overflow shift or leak aborts the process; otherwise it prints how many cases decoded and how many were refused. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int pyr_jpeg_decode(const uint8_t* bytes, size_t nbytes, int* out_width, int* out_height, uint8_t** out_rgb, char* error, size_t error_size);
void pyr_image_free(uint8_t* p);

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    unsigned ok = 0, refused = 0;
    for (;;) {
        uint32_t n;
        if (fread(&n, 4, 1, f) != 1) break;
        /* an exact-size heap copy, so that reading one byte past the file is an ASan error */
        uint8_t* bytes = (uint8_t*)malloc(n ? n : 1);
        if (n && fread(bytes, 1, n, f) != n) return 2;
        int w = 0, h = 0;
        uint8_t* rgb = NULL;
        char error[128] = "";
        if (pyr_jpeg_decode(bytes, n, &w, &h, &rgb, error, sizeof(error)) == 0) {
            volatile uint8_t sink = rgb[(size_t)w * h * 3 - 1]; /* the whole buffer must be there */
            (void)sink;
            pyr_image_free(rgb);
            ++ok;
        } else {
            if (!error[0]) return 3; /* a failure without a message */
            ++refused;
        }
        free(bytes);
    }
    fclose(f);
    printf("%u decoded, %u refused\n", ok, refused);
    return 0;
}
