// host_emit.cpp — the host-only entry points of include/pmp.h: PartitionMat text / binary emission and frame tiling
// (Map2Partition.py:385-412; consumer EncAppCfg::parsePartitionMatrix, EncAppCfg.cpp:4234-4404), plus the context-less
// error string.  No HIP in this file: it is also compiled with -fsanitize=address,undefined into the CPU-only test library
// (make hostasan, tests/test_hostasan_cpu.py).
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "pmp_hostonly.h"

namespace pmp {

static thread_local std::string g_err;

int set_err_global(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

const char *global_err() { return g_err.c_str(); }

}  // namespace pmp

using pmp::set_err_global;
#define set_err(ctx_unused, code, msg) set_err_global((code), (msg))

extern "C" {

// ---- PartitionMat text (Map2Partition.py:385-412) ---------------------------------------------------------
// Emission is the host-side hot spot of the path: 645 k lines per 1080p frame and file.  Values are in
// {-1, 0, 1, 2, 3}, so a line is "d\n" or "-1\n": 16 values of one block row are contiguous in the per-block arrays and are
// expanded with one 16-bit store each (an unconditional '-' is written first and kept only for negative values).
static inline char *emit_u8_row(char *p, const uint8_t *v, int count)
{
    unsigned m = 0;
    for (int i = 0; i < count; ++i) m |= v[i];
    if (m < 8) {   // every value is a single digit: fixed 2 bytes per value, no branches
        for (int i = 0; i < count; ++i) { p[2 * i] = (char)('0' + v[i]); p[2 * i + 1] = '\n'; }
        return p + 2 * count;
    }
    for (int i = 0; i < count; ++i) {
        const unsigned d = v[i];
        if (d < 10) { p[0] = (char)('0' + d); p[1] = '\n'; p += 2; }
        else { p += snprintf(p, 8, "%u\n", d); }   // never produced by the path; kept for arbitrary caller data
    }
    return p;
}

static inline char *emit_i8_row(char *p, const int8_t *v, int count)
{
    for (int i = 0; i < count; ++i) {
        const int d = v[i];
        p[0] = '-';
        p += d < 0;
        const int a = d < 0 ? -d : d;
        if (a < 10) { p[0] = (char)('0' + a); p[1] = '\n'; p += 2; }
        else { p += snprintf(p, 8, "%d\n", a); }
    }
    return p;
}

int64_t pmp_format_partition_text(int frames, int H, int W, const uint8_t *hor, const uint8_t *ver, const uint8_t *qt_u8,
                                  const int8_t *dire, char *buf, int64_t cap)
{
    if (frames < 0 || H < 0 || W < 0 || !hor || !ver || !qt_u8 || !dire) return set_err(nullptr, PMP_E_INVALID, "pmp_format_partition_text: bad arguments");
    const int bh = H / 64, bw = W / 64, R = 16 * bh;
    const int64_t nblk = (int64_t)frames * bh * bw;
    if (!buf) {   // exact size: 2 bytes per value, +1 per negative direction, +digits beyond one for values >= 10
        int64_t need = nblk * (256 + 256 + 64 + 768) * 2, neg = 0;
        int amax = 0;
        for (int64_t i = 0; i < nblk * 768; ++i) { neg += dire[i] < 0; const int a = dire[i] < 0 ? -dire[i] : dire[i]; amax = a > amax ? a : amax; }
        unsigned umax = 0;
        for (int64_t i = 0; i < nblk * 256; ++i) { umax = hor[i] > umax ? hor[i] : umax; umax = ver[i] > umax ? ver[i] : umax; }
        for (int64_t i = 0; i < nblk * 64; ++i) umax = qt_u8[i] > umax ? qt_u8[i] : umax;
        need += neg;
        if (amax >= 10 || umax >= 10) {   // never on the path's own data; exact for arbitrary caller data
            for (int64_t i = 0; i < nblk * 768; ++i) { const int a = dire[i] < 0 ? -dire[i] : dire[i]; need += (a >= 10) + (a >= 100); }
            for (int64_t i = 0; i < nblk * 256; ++i) need += (hor[i] >= 10) + (hor[i] >= 100) + (ver[i] >= 10) + (ver[i] >= 100);
            for (int64_t i = 0; i < nblk * 64; ++i) need += (qt_u8[i] >= 10) + (qt_u8[i] >= 100);
        }
        return need;
    }
    // A row of 16 values is at most 16 * 5 bytes.  Rows are written straight into the buffer while that much room is left;
    // the last rows of an exactly-sized buffer go through a bounce buffer, so `cap == size` is enough and never overrun.
    char *p = buf, *end = buf + cap;
    bool ok = true;
    auto put_u8 = [&](const uint8_t *v, int count) {
        if (end - p >= 80) { p = emit_u8_row(p, v, count); return; }
        char tmp[96];
        const size_t len = (size_t)(emit_u8_row(tmp, v, count) - tmp);
        if ((size_t)(end - p) < len) { ok = false; return; }
        memcpy(p, tmp, len);
        p += len;
    };
    auto put_i8 = [&](const int8_t *v, int count) {
        if (end - p >= 80) { p = emit_i8_row(p, v, count); return; }
        char tmp[96];
        const size_t len = (size_t)(emit_i8_row(tmp, v, count) - tmp);
        if ((size_t)(end - p) < len) { ok = false; return; }
        memcpy(p, tmp, len);
        p += len;
    };
    for (int f = 0; f < frames && ok; ++f) {
        const int64_t base = (int64_t)f * bh * bw;
        for (int plane = 0; plane < 2; ++plane) {
            const uint8_t *src = plane ? ver : hor;
            for (int r = 0; r < R && ok; ++r)
                for (int bx = 0; bx < bw && ok; ++bx) put_u8(src + (base + (int64_t)(r >> 4) * bw + bx) * 256 + (r & 15) * 16, 16);
        }
        for (int r = 0; r < R / 2 && ok; ++r)
            for (int bx = 0; bx < bw && ok; ++bx) put_u8(qt_u8 + (base + (int64_t)(r >> 3) * bw + bx) * 64 + (r & 7) * 8, 8);
        for (int k = 0; k < 3; ++k)
            for (int r = 0; r < R && ok; ++r)
                for (int bx = 0; bx < bw && ok; ++bx)
                    put_i8(dire + (base + (int64_t)(r >> 4) * bw + bx) * 768 + k * 256 + (r & 15) * 16, 16);
    }
    if (!ok) return set_err(nullptr, PMP_E_INVALID, "pmp_format_partition_text: buffer too small");
    return p - buf;
}

int pmp_write_partition_file(const char *path, int frames, int H, int W, const uint8_t *hor, const uint8_t *ver,
                             const uint8_t *qt_u8, const int8_t *dire)
{
    if (!path) return set_err(nullptr, PMP_E_INVALID, "pmp_write_partition_file: null path");
    const int64_t need = pmp_format_partition_text(frames, H, W, hor, ver, qt_u8, dire, nullptr, 0);
    if (need < 0) return (int)need;
    std::unique_ptr<char[]> buf(new (std::nothrow) char[(size_t)need + 1]);
    if (!buf) return set_err(nullptr, PMP_E_NOMEM, "pmp_write_partition_file: out of host memory");
    if (need && pmp_format_partition_text(frames, H, W, hor, ver, qt_u8, dire, buf.get(), need) != need)
        return set_err(nullptr, PMP_E_INVALID, "pmp_write_partition_file: formatting failed");
    FILE *fp = fopen(path, "wb");
    if (!fp) return set_err(nullptr, PMP_E_IO, std::string("cannot open ") + path);
    const size_t wr = need ? fwrite(buf.get(), 1, (size_t)need, fp) : 0;
    const int cl = fclose(fp);
    if (wr != (size_t)need || cl != 0) return set_err(nullptr, PMP_E_IO, std::string("short write to ") + path);
    return PMP_OK;
}

int pmp_write_partition_binary(const char *path, int frames, int H, int W, const uint8_t *hor, const uint8_t *ver,
                               const uint8_t *qt_u8, const int8_t *dire)
{
    if (!path || frames < 0 || H < 0 || W < 0 || !hor || !ver || !qt_u8 || !dire)
        return set_err(nullptr, PMP_E_INVALID, "pmp_write_partition_binary: bad arguments");
    const int bh = H / 64, bw = W / 64, R = 16 * bh, C = 16 * bw;
    FILE *fp = fopen(path, "wb");
    if (!fp) return set_err(nullptr, PMP_E_IO, std::string("cannot open ") + path);
    const char magic[8] = {'P', 'M', 'P', 'B', '1', 0, 0, 0};
    const int32_t hdr[8] = {frames, H, W, R, C, 0, 0, 0};
    bool ok = fwrite(magic, 1, 8, fp) == 8 && fwrite(hdr, 4, 8, fp) == 8;
    std::vector<uint8_t> row((size_t)(C > 0 ? C : 1));
    for (int f = 0; f < frames && ok; ++f) {
        const int64_t base = (int64_t)f * bh * bw;
        for (int plane = 0; plane < 2 && ok; ++plane) {
            const uint8_t *src = plane ? ver : hor;
            for (int r = 0; r < R && ok; ++r) {
                for (int cc = 0; cc < C; ++cc) row[cc] = src[(base + (r >> 4) * bw + (cc >> 4)) * 256 + (r & 15) * 16 + (cc & 15)];
                ok = fwrite(row.data(), 1, (size_t)C, fp) == (size_t)C;
            }
        }
        for (int r = 0; r < R / 2 && ok; ++r) {
            for (int cc = 0; cc < C / 2; ++cc) row[cc] = qt_u8[(base + (r >> 3) * bw + (cc >> 3)) * 64 + (r & 7) * 8 + (cc & 7)];
            ok = fwrite(row.data(), 1, (size_t)(C / 2), fp) == (size_t)(C / 2);
        }
        for (int k = 0; k < 3 && ok; ++k)
            for (int r = 0; r < R && ok; ++r) {
                for (int cc = 0; cc < C; ++cc)
                    row[cc] = (uint8_t)dire[(base + (r >> 4) * bw + (cc >> 4)) * 768 + k * 256 + (r & 15) * 16 + (cc & 15)];
                ok = fwrite(row.data(), 1, (size_t)C, fp) == (size_t)C;
            }
    }
    const int cl = fclose(fp);
    if (!ok || cl != 0) return set_err(nullptr, PMP_E_IO, std::string("short write to ") + path);
    return PMP_OK;
}

int pmp_tile_partition_maps(int frames, int H, int W, const uint8_t *hor, const uint8_t *ver, const uint8_t *qt_u8,
                            const int8_t *dire, uint8_t *out_hor, uint8_t *out_ver, uint8_t *out_qt, int8_t *out_dire)
{
    if (frames < 0 || H < 0 || W < 0 || !hor || !ver || !qt_u8 || !dire || !out_hor || !out_ver || !out_qt || !out_dire)
        return set_err(nullptr, PMP_E_INVALID, "pmp_tile_partition_maps: bad arguments");
    const int bh = H / 64, bw = W / 64, R = 16 * bh, C = 16 * bw;
    for (int f = 0; f < frames; ++f) {
        const int64_t base = (int64_t)f * bh * bw;
        for (int r = 0; r < R; ++r)
            for (int cc = 0; cc < C; ++cc) {
                const int64_t blk = base + (r >> 4) * bw + (cc >> 4);
                const int cell = (r & 15) * 16 + (cc & 15);
                const int64_t o = ((int64_t)f * R + r) * C + cc;
                out_hor[o] = hor[blk * 256 + cell];
                out_ver[o] = ver[blk * 256 + cell];
                for (int k = 0; k < 3; ++k) out_dire[(((int64_t)f * 3 + k) * R + r) * C + cc] = dire[blk * 768 + k * 256 + cell];
            }
        for (int r = 0; r < R / 2; ++r)
            for (int cc = 0; cc < C / 2; ++cc)
                out_qt[((int64_t)f * (R / 2) + r) * (C / 2) + cc] = qt_u8[(base + (r >> 3) * bw + (cc >> 3)) * 64 + (r & 7) * 8 + (cc & 7)];
    }
    return PMP_OK;
}

}  // extern "C"
