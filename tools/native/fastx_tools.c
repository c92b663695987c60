/* Helpers of the throughput tools (not part of the product library): write synthetic reads as FASTQ at memory speed and
 * compress a file to BGZF with all cores, so that a 12.5 M-read input (25 GB) is on disk within a minute.
 *   gcc -O2 -fopenmp -shared -fPIC -o libfastx_tools.so fastx_tools.c -lz                                             */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* appends n records "@<prefix><first_id + i>\n<bases>\n+\n<I...>\n" to path; returns bytes written, < 0 on error */
int64_t fq_append(const char* path, const uint8_t* bases, const uint64_t* off, uint64_t n, uint64_t first_id, const char* prefix)
{
    FILE* f = fopen(path, "ab");
    if (!f) return -1;
    size_t cap = 64u << 20, used = 0;
    char* buf = (char*)malloc(cap + (32u << 20));
    if (!buf) { fclose(f); return -2; }
    int64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t L = off[i + 1] - off[i];
        if (used + 2 * L + 128 > cap + (32u << 20) || used > cap) {
            if (fwrite(buf, 1, used, f) != used) { free(buf); fclose(f); return -3; }
            total += (int64_t)used; used = 0;
        }
        used += (size_t)sprintf(buf + used, "@%s%llu\n", prefix, (unsigned long long)(first_id + i));
        memcpy(buf + used, bases + off[i], L); used += L;
        buf[used++] = '\n'; buf[used++] = '+'; buf[used++] = '\n';
        memset(buf + used, 'I', L); used += L;
        buf[used++] = '\n';
    }
    if (used && fwrite(buf, 1, used, f) != used) { free(buf); fclose(f); return -3; }
    total += (int64_t)used;
    free(buf);
    if (fclose(f) != 0) return -4;
    return total;
}

/* appends n unmapped records (flag 4, quality 0xFF = absent) to the UNCOMPRESSED BAM at path, writing the header first when
 * first_id == 0 (SAM specification 4.2; bgzf_compress_file turns the result into a .bam); returns bytes written */
int64_t bam_raw_append(const char* path, const uint8_t* bases, const uint64_t* off, uint64_t n, uint64_t first_id, const char* prefix)
{
    FILE* f = fopen(path, first_id == 0 ? "wb" : "ab");
    if (!f) return -1;
    static uint8_t code[256];
    memset(code, 15, sizeof(code));
    code['A'] = 1; code['C'] = 2; code['G'] = 4; code['T'] = 8; code['N'] = 15;
    size_t cap = 64u << 20, used = 0;
    uint8_t* buf = (uint8_t*)malloc(cap + (32u << 20));
    if (!buf) { fclose(f); return -2; }
    int64_t total = 0;
    if (first_id == 0) {
        static const char text[] = "@HD\tVN:1.6\tSO:unsorted\n";
        const uint32_t l_text = (uint32_t)(sizeof(text) - 1), n_ref = 0;
        memcpy(buf, "BAM\1", 4); memcpy(buf + 4, &l_text, 4); memcpy(buf + 8, text, l_text); memcpy(buf + 8 + l_text, &n_ref, 4);
        used = 12 + l_text;
    }
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t L = off[i + 1] - off[i];
        if (used + 2 * L + 256 > cap + (32u << 20) || used > cap) {
            if (fwrite(buf, 1, used, f) != used) { free(buf); fclose(f); return -3; }
            total += (int64_t)used; used = 0;
        }
        char name[64];
        const uint32_t l_name = (uint32_t)sprintf(name, "%s%llu", prefix, (unsigned long long)(first_id + i)) + 1;
        const uint32_t l_seq = (uint32_t)L, block = 32 + l_name + (l_seq + 1) / 2 + l_seq;
        uint8_t* o = buf + used;
        const int32_t minus1 = -1; const uint32_t zero = 0;
        memcpy(o, &block, 4); memcpy(o + 4, &minus1, 4); memcpy(o + 8, &minus1, 4);
        o[12] = (uint8_t)l_name; o[13] = 0; o[14] = 0x48; o[15] = 0x12;                 /* l_read_name, mapq, bin 4680 */
        o[16] = 0; o[17] = 0; o[18] = 4; o[19] = 0;                                     /* n_cigar_op 0, flag 4 */
        memcpy(o + 20, &l_seq, 4); memcpy(o + 24, &minus1, 4); memcpy(o + 28, &minus1, 4); memcpy(o + 32, &zero, 4);
        memcpy(o + 36, name, l_name);
        uint8_t* q = o + 36 + l_name;
        const uint8_t* b = bases + off[i];
        for (uint64_t k = 0; k + 1 < L; k += 2) *q++ = (uint8_t)(code[b[k]] << 4 | code[b[k + 1]]);
        if (L & 1) *q++ = (uint8_t)(code[b[L - 1]] << 4);
        memset(q, 0xFF, L);
        used += 4 + block;
    }
    if (used && fwrite(buf, 1, used, f) != used) { free(buf); fclose(f); return -3; }
    total += (int64_t)used;
    free(buf);
    if (fclose(f) != 0) return -4;
    return total;
}

/* in -> out as BGZF (SAM specification 4.1: gzip members of <= 64 KiB with the 'BC' size field, plus the end marker);
 * blocks are compressed in parallel, written in order; returns bytes written, < 0 on error */
int64_t bgzf_compress_file(const char* in, const char* out, int level)
{
    FILE* fi = fopen(in, "rb");
    FILE* fo = fopen(out, "wb");
    if (!fi || !fo) { if (fi) fclose(fi); if (fo) fclose(fo); return -1; }
    enum { BLK = 65280, NB = 4096, OB = 66000 };
    uint8_t* ibuf = (uint8_t*)malloc((size_t)BLK * NB);
    uint8_t* obuf = (uint8_t*)malloc((size_t)OB * NB);
    uint32_t* olen = (uint32_t*)malloc(sizeof(uint32_t) * NB);
    int64_t total = 0; int bad = 0;
    for (;;) {
        const size_t got = fread(ibuf, 1, (size_t)BLK * NB, fi);
        if (got == 0) break;
        const long nb = (long)((got + BLK - 1) / BLK);
#pragma omp parallel for schedule(dynamic, 8)
        for (long b = 0; b < nb; ++b) {
            const size_t a = (size_t)b * BLK, len = a + BLK <= got ? BLK : got - a;
            uint8_t* o = obuf + (size_t)b * OB;
            z_stream z; memset(&z, 0, sizeof(z));
            if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; continue; }
            z.next_in = ibuf + a; z.avail_in = (uInt)len; z.next_out = o + 18; z.avail_out = OB - 18 - 8;
            if (deflate(&z, Z_FINISH) != Z_STREAM_END) bad = 1;
            const uint32_t clen = (uint32_t)(OB - 18 - 8 - z.avail_out), tot = 18 + clen + 8;
            deflateEnd(&z);
            static const uint8_t hdr[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
            memcpy(o, hdr, 16);
            o[16] = (uint8_t)((tot - 1) & 0xff); o[17] = (uint8_t)((tot - 1) >> 8);
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), ibuf + a, (uInt)len);
            uint8_t* t = o + 18 + clen;
            t[0] = crc & 0xff; t[1] = (crc >> 8) & 0xff; t[2] = (crc >> 16) & 0xff; t[3] = (uint8_t)(crc >> 24);
            t[4] = len & 0xff; t[5] = (len >> 8) & 0xff; t[6] = 0; t[7] = 0;
            olen[b] = tot;
        }
        if (bad) break;
        for (long b = 0; b < nb; ++b) {
            if (fwrite(obuf + (size_t)b * OB, 1, olen[b], fo) != olen[b]) { bad = 1; break; }
            total += olen[b];
        }
        if (bad) break;
    }
    static const uint8_t eof_marker[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    if (!bad && fwrite(eof_marker, 1, 28, fo) != 28) bad = 1;
    total += 28;
    free(ibuf); free(obuf); free(olen);
    fclose(fi);
    if (fclose(fo) != 0) bad = 1;
    return bad ? -2 : total;
}
