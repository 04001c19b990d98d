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

}  // extern "C" (reopened below)

namespace {

// Per-block arrays seen through byte strides: the four dense arrays of the classic entry points (strides 256/256/64/768) or the
// fields of packed 1344-byte records (PMP_RECORD_BYTES: hor | ver | qt | dire), which is what the multi-GPU path holds.
struct BlockView {
    const uint8_t *hor, *ver, *qt;
    const int8_t *dire;
    int64_t sh, sv, sq, sd;
};

inline int64_t u8_len(const uint8_t *v, int n)
{
    int64_t len = 2 * n;
    for (int i = 0; i < n; ++i) len += (v[i] >= 10) + (v[i] >= 100);
    return len;
}

inline int64_t i8_len(const int8_t *v, int n)
{
    int64_t len = 2 * n;
    for (int i = 0; i < n; ++i) { const int d = v[i], a = d < 0 ? -d : d; len += (d < 0) + (a >= 10) + (a >= 100); }
    return len;
}

// Exact text size of the six sections (hor, ver, qt, dire 0..2) of ONE block row of `bw` blocks starting at block `b0`.
void row_section_sizes(const BlockView &v, int64_t b0, int bw, int64_t out[6])
{
    for (int k = 0; k < 6; ++k) out[k] = 0;
    for (int bx = 0; bx < bw; ++bx) {
        const int64_t b = b0 + bx;
        out[0] += u8_len(v.hor + b * v.sh, 256);
        out[1] += u8_len(v.ver + b * v.sv, 256);
        out[2] += u8_len(v.qt + b * v.sq, 64);
        for (int k = 0; k < 3; ++k) out[3 + k] += i8_len(v.dire + b * v.sd + k * 256, 256);
    }
}

// Text of `nbr` consecutive block rows (bw blocks each, first block b0) in the file's order INSIDE a frame - section-major: all hor
// matrix rows, all ver rows, the qt rows, then the three dire matrices (Map2Partition.py:401-412).  A whole frame is nbr = H/64.
// A row of 16 values is at most 16 * 5 bytes.  Rows are written straight into the buffer while that much room is left; the last
// rows of an exactly-sized buffer go through a bounce buffer, so `cap == size` is enough and never overrun.
bool format_rows(const BlockView &v, int64_t b0, int nbr, int bw, char *&p, char *end)
{
    bool ok = true;
    auto put_u8 = [&](const uint8_t *src, int count) {
        if (end - p >= 80) { p = emit_u8_row(p, src, count); return; }
        char tmp[96];
        const size_t len = (size_t)(emit_u8_row(tmp, src, count) - tmp);
        if ((size_t)(end - p) < len) { ok = false; return; }
        memcpy(p, tmp, len);
        p += len;
    };
    auto put_i8 = [&](const int8_t *src, int count) {
        if (end - p >= 80) { p = emit_i8_row(p, src, count); return; }
        char tmp[96];
        const size_t len = (size_t)(emit_i8_row(tmp, src, count) - tmp);
        if ((size_t)(end - p) < len) { ok = false; return; }
        memcpy(p, tmp, len);
        p += len;
    };
    const int R = 16 * nbr;
    for (int plane = 0; plane < 2; ++plane) {
        const uint8_t *src = plane ? v.ver : v.hor;
        const int64_t st = plane ? v.sv : v.sh;
        for (int r = 0; r < R && ok; ++r)
            for (int bx = 0; bx < bw && ok; ++bx) put_u8(src + (b0 + (int64_t)(r >> 4) * bw + bx) * st + (r & 15) * 16, 16);
    }
    for (int r = 0; r < R / 2 && ok; ++r)
        for (int bx = 0; bx < bw && ok; ++bx) put_u8(v.qt + (b0 + (int64_t)(r >> 3) * bw + bx) * v.sq + (r & 7) * 8, 8);
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < R && ok; ++r)
            for (int bx = 0; bx < bw && ok; ++bx) put_i8(v.dire + (b0 + (int64_t)(r >> 4) * bw + bx) * v.sd + k * 256 + (r & 15) * 16, 16);
    return ok;
}

int64_t format_block_rows(const char *who, int W, int nbr, const BlockView &v, char *buf, int64_t cap, int64_t *row_bytes)
{
    if (W < 0 || nbr < 0 || !v.hor || !v.ver || !v.qt || !v.dire) return set_err_global(PMP_E_INVALID, std::string(who) + ": bad arguments");
    const int bw = W / 64;
    int64_t total = 0;
    if (row_bytes || !buf) {
        for (int r = 0; r < nbr; ++r) {
            int64_t sz[6];
            row_section_sizes(v, (int64_t)r * bw, bw, sz);
            for (int k = 0; k < 6; ++k) { total += sz[k]; if (row_bytes) row_bytes[r * 6 + k] = sz[k]; }
        }
        if (!buf) return total;
    }
    char *p = buf;
    if (!format_rows(v, 0, nbr, bw, p, buf + cap)) return set_err_global(PMP_E_INVALID, std::string(who) + ": buffer too small");
    return p - buf;
}

}  // namespace

extern "C" {

int64_t pmp_format_partition_text(int frames, int H, int W, const uint8_t *hor, const uint8_t *ver, const uint8_t *qt_u8,
                                  const int8_t *dire, char *buf, int64_t cap)
{
    if (frames < 0 || H < 0 || W < 0 || !hor || !ver || !qt_u8 || !dire) return set_err(nullptr, PMP_E_INVALID, "pmp_format_partition_text: bad arguments");
    const int bh = H / 64, bw = W / 64;
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
    const BlockView v{hor, ver, qt_u8, dire, 256, 256, 64, 768};
    char *p = buf, *end = buf + cap;
    for (int f = 0; f < frames; ++f)
        if (!format_rows(v, (int64_t)f * bh * bw, bh, bw, p, end)) return set_err(nullptr, PMP_E_INVALID, "pmp_format_partition_text: buffer too small");
    return p - buf;
}

int64_t pmp_format_partition_rows(int W, int block_rows, const uint8_t *hor, const uint8_t *ver, const uint8_t *qt_u8, const int8_t *dire,
                                  char *buf, int64_t cap, int64_t *row_section_bytes)
{
    const BlockView v{hor, ver, qt_u8, dire, 256, 256, 64, 768};
    return format_block_rows("pmp_format_partition_rows", W, block_rows, v, buf, cap, row_section_bytes);
}

int64_t pmp_format_partition_rows_records(int W, int block_rows, const uint8_t *rec, char *buf, int64_t cap, int64_t *row_section_bytes)
{
    const BlockView v{rec, rec ? rec + 256 : nullptr, rec ? rec + 512 : nullptr, rec ? reinterpret_cast<const int8_t *>(rec + 576) : nullptr,
                      PMP_RECORD_BYTES, PMP_RECORD_BYTES, PMP_RECORD_BYTES, PMP_RECORD_BYTES};
    return format_block_rows("pmp_format_partition_rows_records", W, block_rows, v, buf, cap, row_section_bytes);
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

}  // extern "C" (reopened below)

namespace {

// Frame matrices of `nbr` block rows (first block b0): hor, ver u8[16 nbr][C], qt u8[8 nbr][C/2], dire i8[3][16 nbr][C].
void tile_rows(const BlockView &v, int64_t b0, int nbr, int bw, uint8_t *oh, uint8_t *ov, uint8_t *oq, int8_t *od)
{
    const int R = 16 * nbr, C = 16 * bw;
    for (int r = 0; r < R; ++r)
        for (int bx = 0; bx < bw; ++bx) {
            const int64_t blk = b0 + (int64_t)(r >> 4) * bw + bx;
            const int cell = (r & 15) * 16;
            const int64_t o = (int64_t)r * C + bx * 16;
            memcpy(oh + o, v.hor + blk * v.sh + cell, 16);
            memcpy(ov + o, v.ver + blk * v.sv + cell, 16);
            for (int k = 0; k < 3; ++k) memcpy(od + (int64_t)k * R * C + o, v.dire + blk * v.sd + k * 256 + cell, 16);
        }
    for (int r = 0; r < R / 2; ++r)
        for (int bx = 0; bx < bw; ++bx)
            memcpy(oq + (int64_t)r * (C / 2) + bx * 8, v.qt + (b0 + (int64_t)(r >> 3) * bw + bx) * v.sq + (r & 7) * 8, 8);
}

}  // namespace

extern "C" {

int pmp_tile_partition_maps(int frames, int H, int W, const uint8_t *hor, const uint8_t *ver, const uint8_t *qt_u8,
                            const int8_t *dire, uint8_t *out_hor, uint8_t *out_ver, uint8_t *out_qt, int8_t *out_dire)
{
    if (frames < 0 || H < 0 || W < 0 || !hor || !ver || !qt_u8 || !dire || !out_hor || !out_ver || !out_qt || !out_dire)
        return set_err(nullptr, PMP_E_INVALID, "pmp_tile_partition_maps: bad arguments");
    const int bh = H / 64, bw = W / 64;
    const int64_t R = 16 * bh, C = 16 * bw;
    const BlockView v{hor, ver, qt_u8, dire, 256, 256, 64, 768};
    for (int f = 0; f < frames; ++f)
        tile_rows(v, (int64_t)f * bh * bw, bh, bw, out_hor + f * R * C, out_ver + f * R * C, out_qt + f * (R / 2) * (C / 2), out_dire + f * 3 * R * C);
    return PMP_OK;
}

int pmp_tile_partition_rows_records(int W, int block_rows, const uint8_t *rec, uint8_t *out_hor, uint8_t *out_ver, uint8_t *out_qt, int8_t *out_dire)
{
    if (W < 0 || block_rows < 0 || !rec || !out_hor || !out_ver || !out_qt || !out_dire)
        return set_err(nullptr, PMP_E_INVALID, "pmp_tile_partition_rows_records: bad arguments");
    const BlockView v{rec, rec + 256, rec + 512, reinterpret_cast<const int8_t *>(rec + 576), PMP_RECORD_BYTES, PMP_RECORD_BYTES, PMP_RECORD_BYTES,
                      PMP_RECORD_BYTES};
    tile_rows(v, 0, block_rows, W / 64, out_hor, out_ver, out_qt, out_dire);
    return PMP_OK;
}

}  // extern "C"
