// pmp_hook.cpp — consumer side of SURVEY.md 8(f) rows N2 and N4 for the reference's patched VTM-10.0.
//
// Replaces EncAppCfg::parsePartitionMatrix (App/EncoderApp/EncAppCfg.cpp:4234-4404, called from encmain.cpp:184-188): instead
// of 645 k getline + std::stoi calls per 1080p frame and file, the partition matrices come from
//   N2  the binary side channel written by pmp_write_partition_binary / the driver's --binary flag
//       (./PartitionMat/<seq>_{Luma,Chroma}_QP<qp>_PartitionMat.pmpb, layout in include/pmp.h): the files are mmap'ed and the
//       row-pointer tables the encoder indexes (Lib/CommonLib/Rom.h:240-248) point straight into the mappings - zero copy;
//   N4  or, when no such file exists and the hook was built with -DPMP_HOOK_INPROCESS, from libpmp_hip.so in-process: the hook
//       reads the frames the encoder is going to code, runs cutter + nets + Map2Partition on the GPU
//       (pmp_cut_blocks / pmp_infer_postprocess) and tiles the flags with pmp_tile_partition_maps - no file hop at all.
//       Weights: <Comp>_{Q,BD}_<qp>.pmpw under $PMP_MODEL_DIR (default ./CTU_Models, the reference's place, Inference_QBD.py:219-220).
// Like the reference's parser it ends the process when it cannot deliver (EncAppCfg.cpp:4252-4263).
//
// Built by tools/vtm_build/CMakeLists.txt (-DPMP_HOOK=ON) into EncoderAppHook: the reference's App/EncoderApp sources, with
// EncAppCfg.cpp compiled under -DparsePartitionMatrix=parsePartitionMatrix_text so that this definition is the one
// encmain.cpp calls.  No reference source is modified or copied.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "EncAppCfg.h"
#include "CommonLib/Rom.h"
#include "pmp.h"

namespace {

struct Maps { const uint8_t *hor, *ver, *qt; const int8_t *dire; };   // frame matrices of one component: [F][R][C], [F][R/2][C/2], [F][3][R][C]

[[noreturn]] void die(const std::string &msg)
{
    std::cerr << "pmp_hook: " << msg << std::endl;
    exit(1);
}

// mmap a .pmpb file and check its header against the geometry the encoder derived.  Returns the per-frame base and stride.
const uint8_t *map_pmpb(const std::string &path, int frames, int rows, int cols, size_t &per_frame)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0) die("cannot stat " + path);
    per_frame = (size_t)5 * rows * cols + (size_t)rows * cols / 4;
    if ((size_t)st.st_size < 40 + per_frame * frames) die(path + ": shorter than FramesToBeEncoded needs");
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0);   // private: the encoder's tables are non-const
    close(fd);
    if (m == MAP_FAILED) die("mmap failed for " + path);
    const uint8_t *b = static_cast<const uint8_t *>(m);
    int32_t hdr[8];
    memcpy(hdr, b + 8, sizeof(hdr));
    if (memcmp(b, "PMPB1\0\0\0", 8) != 0 || hdr[3] != rows || hdr[4] != cols || hdr[0] < frames)
        die(path + ": not a PMPB1 file of this geometry");
    return b + 40;
}

// Point the encoder's tables (allocated here with the shapes EncAppCfg.cpp:4265-4296 uses) at frame matrices.
void publish(const Maps m[2], int frames, int rows, int cols, const size_t frame_stride[2][4])
{
    partitionHorMat = new uint8_t ***[frames];
    partitionVerMat = new uint8_t ***[frames];
    qtDepthMat = new uint8_t ***[frames];
    directionMat = new int8_t ****[frames];
    for (int f = 0; f < frames; f++) {
        partitionHorMat[f] = new uint8_t **[2];
        partitionVerMat[f] = new uint8_t **[2];
        qtDepthMat[f] = new uint8_t **[2];
        directionMat[f] = new int8_t ***[2];
        for (int k = 0; k < 2; k++) {
            uint8_t *hor = const_cast<uint8_t *>(m[k].hor) + (size_t)f * frame_stride[k][0];
            uint8_t *ver = const_cast<uint8_t *>(m[k].ver) + (size_t)f * frame_stride[k][1];
            uint8_t *qt = const_cast<uint8_t *>(m[k].qt) + (size_t)f * frame_stride[k][2];
            int8_t *dire = const_cast<int8_t *>(m[k].dire) + (size_t)f * frame_stride[k][3];
            partitionHorMat[f][k] = new uint8_t *[rows];
            partitionVerMat[f][k] = new uint8_t *[rows];
            qtDepthMat[f][k] = new uint8_t *[rows >> 1];
            directionMat[f][k] = new int8_t **[3];
            for (int i = 0; i < rows; i++) {
                partitionHorMat[f][k][i] = hor + (size_t)i * cols;
                partitionVerMat[f][k][i] = ver + (size_t)i * cols;
            }
            for (int i = 0; i < (rows >> 1); i++) qtDepthMat[f][k][i] = qt + (size_t)i * (cols >> 1);
            for (int d = 0; d < 3; d++) {
                directionMat[f][k][d] = new int8_t *[rows];
                for (int i = 0; i < rows; i++) directionMat[f][k][d][i] = dire + ((size_t)d * rows + i) * cols;
            }
        }
    }
}

#ifdef PMP_HOOK_INPROCESS
void ck(int rc, pmp_ctx *ctx, const char *what)
{
    if (rc < 0) die(std::string(what) + ": " + pmp_last_error(ctx));
}

// every `ratio`-th frame from `skip` on, planar 4:2:0, 8-bit or 16-bit samples (Inference_QBD.py:78-102)
void read_frames(const std::string &path, int w, int h, int frames, int skip, int ratio, bool wide, std::vector<uint8_t> &y,
                 std::vector<uint8_t> &u, std::vector<uint8_t> &v)
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) die("cannot open " + path);
    const size_t bps = wide ? 2 : 1, ny = (size_t)w * h * bps, nc = ny / 4, fr = ny + 2 * nc;
    y.resize(ny * frames); u.resize(nc * frames); v.resize(nc * frames);
    for (int f = 0; f < frames; f++) {
        if (fseeko(fp, (off_t)((size_t)(skip + f * ratio) * fr), SEEK_SET) != 0 || fread(&y[f * ny], 1, ny, fp) != ny ||
            fread(&u[f * nc], 1, nc, fp) != nc || fread(&v[f * nc], 1, nc, fp) != nc)
            die(path + ": too short for FramesToBeEncoded");
    }
    fclose(fp);
}
#endif

}  // namespace

bool EncAppCfg::parsePartitionMatrix(int argc, char *argv[], int32_t &partitionRow, int32_t &partitionColumn, int32_t &partitionFrameNum)
{
    // sequence name and geometry exactly as the reference derives them (EncAppCfg.cpp:4235-4250)
    std::string seq = m_inputFileName;
    const size_t pos = seq.find_last_of('/');
    seq = seq.substr(pos == std::string::npos ? 0 : pos + 1);
    if (seq.size() > 4) seq = seq.substr(0, seq.size() - 4);
    partitionFrameNum = m_framesToBeEncoded;
    const int ch = (m_iSourceHeight >> 6) * 64, cw = (m_iSourceWidth >> 6) * 64;
    partitionRow = ch >> 2;
    partitionColumn = cw >> 2;
    const int F = partitionFrameNum, R = partitionRow, Cc = partitionColumn;
    const std::string base = "./PartitionMat/" + seq;
    const std::string tail = "_QP" + std::to_string(m_iQP) + "_PartitionMat.pmpb";

    Maps maps[2];
    size_t stride[2][4];
    // ---- N2: binary side channel, mmap'ed
    size_t per = 0;
    const uint8_t *bl = map_pmpb(base + "_Luma" + tail, F, R, Cc, per);
    const uint8_t *bc = bl ? map_pmpb(base + "_Chroma" + tail, F, R, Cc, per) : nullptr;
    if (bl && bc) {
        const uint8_t *b[2] = {bl, bc};
        for (int k = 0; k < 2; k++) {
            maps[k].hor = b[k];
            maps[k].ver = b[k] + (size_t)R * Cc;
            maps[k].qt = b[k] + (size_t)2 * R * Cc;
            maps[k].dire = reinterpret_cast<const int8_t *>(b[k] + (size_t)2 * R * Cc + (size_t)R * Cc / 4);
            for (int j = 0; j < 4; j++) stride[k][j] = per;
        }
        publish(maps, F, R, Cc, stride);
        std::cout << "pmp_hook: partition maps mmap'ed from " << base << "_{Luma,Chroma}" << tail << std::endl;
        return true;
    }
#ifdef PMP_HOOK_INPROCESS
    // ---- N4: run the prediction path in this process
    {
        const char *md = getenv("PMP_MODEL_DIR");
        const std::string model_dir = md ? md : "./CTU_Models";
        const bool wide = m_inputBitDepth[0] > 8;
        pmp_ctx *ctx = nullptr;
        ck(pmp_create(0, &ctx), nullptr, "pmp_create");
        std::vector<uint8_t> y, u, v;
        read_frames(m_inputFileName, m_iSourceWidth, m_iSourceHeight, F, (int)m_FrameSkip, (int)m_temporalSubsampleRatio, wide, y, u, v);
        const int64_t n = (int64_t)F * (ch / 64) * (cw / 64);
        std::vector<uint8_t> by((size_t)n * 68 * 68), bu((size_t)n * 34 * 34), bv((size_t)n * 34 * 34);
        ck(pmp_cut_blocks(ctx, y.data(), u.data(), v.data(), F, m_iSourceHeight, m_iSourceWidth, wide ? 10 : 8, by.data(), bu.data(), bv.data()),
           ctx, "pmp_cut_blocks");
        static std::vector<uint8_t> own_u8[2][3];       // the encoder keeps pointers into these for its whole run
        static std::vector<int8_t> own_i8[2];
        std::vector<uint8_t> hor((size_t)n * 256), ver((size_t)n * 256), qt((size_t)n * 64);
        std::vector<int8_t> dire((size_t)n * 768);
        for (int k = 0; k < 2; k++) {
            const std::string comp = k ? "Chroma" : "Luma";
            const std::string wq = model_dir + "/" + comp + "_Q_" + std::to_string(m_iQP) + ".pmpw";
            const std::string wb = model_dir + "/" + comp + "_BD_" + std::to_string(m_iQP) + ".pmpw";
            ck(pmp_load_weights_file(ctx, k ? PMP_NET_CHROMA_Q : PMP_NET_LUMA_Q, m_iQP, wq.c_str()), ctx, "QT-net weights");
            ck(pmp_load_weights_file(ctx, k ? PMP_NET_CHROMA_MSBD : PMP_NET_LUMA_MSBD, m_iQP, wb.c_str()), ctx, "MTT-net weights");
            ck(pmp_infer_postprocess(ctx, k ? PMP_CHROMA : PMP_LUMA, m_iQP, by.data(), bu.data(), bv.data(), n, hor.data(), ver.data(), qt.data(),
                                     dire.data(), nullptr, nullptr, nullptr),
               ctx, "pmp_infer_postprocess");
            own_u8[k][0].resize((size_t)F * R * Cc); own_u8[k][1].resize((size_t)F * R * Cc); own_u8[k][2].resize((size_t)F * R * Cc / 4);
            own_i8[k].resize((size_t)F * 3 * R * Cc);
            ck(pmp_tile_partition_maps(F, m_iSourceHeight, m_iSourceWidth, hor.data(), ver.data(), qt.data(), dire.data(), own_u8[k][0].data(),
                                       own_u8[k][1].data(), own_u8[k][2].data(), own_i8[k].data()),
               nullptr, "pmp_tile_partition_maps");
            maps[k].hor = own_u8[k][0].data(); maps[k].ver = own_u8[k][1].data(); maps[k].qt = own_u8[k][2].data(); maps[k].dire = own_i8[k].data();
            stride[k][0] = stride[k][1] = (size_t)R * Cc; stride[k][2] = (size_t)R * Cc / 4; stride[k][3] = (size_t)3 * R * Cc;
        }
        if (pmp_get_saturation(ctx) > 0) std::cerr << "pmp_hook: f16x3 range guard fired; the pass was re-run on bf16x6" << std::endl;
        pmp_destroy(ctx);
        publish(maps, F, R, Cc, stride);
        std::cout << "pmp_hook: partition maps predicted in-process by " << pmp_version() << " (" << n << " blocks per component)" << std::endl;
        return true;
    }
#else
    die("cannot open " + base + "_{Luma,Chroma}" + tail + " (binary side channel; built without PMP_HOOK_INPROCESS)");
#endif
}
