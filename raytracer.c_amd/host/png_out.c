/* png_out.c -- PNG output with the stb_image_write call signature.
 *
 * The reference writes its image with stbi_write_png() from the vendored
 * single-header stb_image_write (lib/stb_image_write.h, called at main.c:41).
 * That header is third-party code we do not copy; this file provides the one
 * entry point the reference's caller uses, with the same name, arguments and
 * return convention (non-zero on success), so main.c-style callers link
 * unchanged.  Encoding: 8-bit, `comp` channels (1 grey, 2 grey+alpha, 3 RGB,
 * 4 RGBA), filter type 0 on every scanline, one zlib stream from the system
 * zlib (deflate level 6).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

int stbi_write_png(char const *filename, int w, int h, int comp, const void *data, int stride_in_bytes);

static void be32(uint8_t *p, uint32_t v)
{
  p[0] = (uint8_t)(v >> 24);
  p[1] = (uint8_t)(v >> 16);
  p[2] = (uint8_t)(v >> 8);
  p[3] = (uint8_t)v;
}

static int chunk(FILE *f, const char type[4], const uint8_t *body, uint32_t len)
{
  uint8_t head[8], tail[4];
  be32(head, len);
  memcpy(head + 4, type, 4);
  uLong crc = crc32(0L, head + 4, 4);
  if (len)
    crc = crc32(crc, body, len);
  be32(tail, (uint32_t)crc);
  return fwrite(head, 1, 8, f) == 8 && (len == 0 || fwrite(body, 1, len, f) == len) && fwrite(tail, 1, 4, f) == 4;
}

int stbi_write_png(char const *filename, int w, int h, int comp, const void *data, int stride_in_bytes)
{
  static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  static const uint8_t colour_type[5] = {0, 0, 4, 2, 6};
  if (!filename || !data || w <= 0 || h <= 0 || comp < 1 || comp > 4)
    return 0;
  if (stride_in_bytes == 0)
    stride_in_bytes = w * comp;

  const size_t row = (size_t)w * (size_t)comp;
  const size_t raw_len = (row + 1) * (size_t)h;
  uint8_t *raw = (uint8_t *)malloc(raw_len);
  if (!raw)
    return 0;
  for (int y = 0; y < h; y++)
  {
    raw[(row + 1) * (size_t)y] = 0; /* filter: none */
    memcpy(raw + (row + 1) * (size_t)y + 1, (const uint8_t *)data + (size_t)stride_in_bytes * (size_t)y, row);
  }
  uLongf z_len = compressBound((uLong)raw_len);
  uint8_t *z = (uint8_t *)malloc(z_len);
  if (!z || compress2(z, &z_len, raw, (uLong)raw_len, 6) != Z_OK)
  {
    free(raw);
    free(z);
    return 0;
  }
  free(raw);

  FILE *f = fopen(filename, "wb");
  if (!f)
  {
    free(z);
    return 0;
  }
  uint8_t ihdr[13];
  be32(ihdr, (uint32_t)w);
  be32(ihdr + 4, (uint32_t)h);
  ihdr[8] = 8;
  ihdr[9] = colour_type[comp];
  ihdr[10] = 0;
  ihdr[11] = 0;
  ihdr[12] = 0;
  int ok = fwrite(signature, 1, 8, f) == 8 && chunk(f, "IHDR", ihdr, 13) && chunk(f, "IDAT", z, (uint32_t)z_len) &&
           chunk(f, "IEND", NULL, 0);
  ok = (fclose(f) == 0) && ok;
  free(z);
  return ok ? 1 : 0;
}
