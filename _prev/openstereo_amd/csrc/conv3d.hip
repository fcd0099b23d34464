// 3-D cost-aggregation convolutions for gfx950: host side (tile selection, LDS geometry, weight packing, C ABI) and the
// instantiations of the stage -> barrier -> taps kernel.  Kernel template: conv_kernel.h.
#include "conv_kernel.h"
#include "conv_inst.h"

namespace osa {

// ------------------------------------------------------------------ dispatch --
// Geometry of a tile configuration; the kernels themselves live in per-mode tables (conv_inst.h: one translation unit per arithmetic
// mode, compiled in parallel).
struct KernelCfg {
    const char* name;
    int M, N;              // voxels / channels per workgroup
    int TD, TH, TW, threads;
    int ks;                // split-K groups inside the workgroup (0 / 1: none)
    int table, idx;        // 0: ConvFnTables::cfgs[idx], 1: ::ks[idx], 2: ::deconv[idx]
};

static const KernelCfg g_cfgs[] = {
#define OSA_CFG_X(RING, MT, NT, WM, WN, TH, TW)                                                                  \
    { #MT "x" #NT "_" #WM "x" #WN "_" #TH "x" #TW, WM * MT * 32, WN * NT * 32, WM * MT * 32 / (TH * TW), TH, TW, WM * WN * 64, 0, 0, -1 },
#define OSA_KS_X(MT, NT, WM, WN, TH, TW, KS)
#include "conv_cfgs.def"
#undef OSA_CFG_X
#undef OSA_KS_X
};
constexpr int N_CFGS = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

// split-K tiles for small maps (conv_kernel.h, KS): KS groups of WM x WN waves share a workgroup, each walks 1 / KS of the input chunks
static const KernelCfg g_ks_cfgs[] = {
#define OSA_CFG_X(RING, MT, NT, WM, WN, TH, TW)
#define OSA_KS_X(MT, NT, WM, WN, TH, TW, KS)                                                                     \
    { #MT "x" #NT "_" #WM "x" #WN "_" #TH "x" #TW "_ks" #KS, WM * MT * 32, WN * NT * 32, WM * MT * 32 / (TH * TW), TH, TW, WM * WN * KS * 64, KS, 1, -1 },
#include "conv_cfgs.def"
#undef OSA_CFG_X
#undef OSA_KS_X
};

// fused transposed conv: 128 input-resolution positions x 32 channels x 8 parity classes per workgroup
static const KernelCfg g_deconv_redir_cfg = { "deconv8_redir_1x1_4x1_4x8", 128, 32, 4, 4, 8, 256, 0, 2, 0 };
static const KernelCfg g_deconv_redir64_cfg = { "deconv8_redir64_1x1_4x1_4x8", 128, 32, 4, 4, 8, 256, 0, 2, 1 };
static const KernelCfg g_deconv_cfg = { "deconv8_1x1_4x1_4x8", 128, 32, 4, 4, 8, 256, 0, 2, 2 };
// fused 2-D transposed conv (D = 1): 128 input-resolution pixels x 32 channels x 4 parity classes
static const KernelCfg g_deconv_flat_cfg = { "deconv4_1x1_4x1_8x16", 128, 32, 1, 8, 16, 256, 0, 2, 3 };

static const KernelFns& fns_of(const KernelCfg& k, int prec) {
    const ConvFnTables& t = (prec == PREC_F32) ? conv_tables_f32() : ((prec == PREC_F16X3) ? conv_tables_f16x3() : conv_tables_f16());
    if (k.table == 0) return t.cfgs[&k - g_cfgs];
    if (k.table == 1) return t.ks[&k - g_ks_cfgs];
    return t.deconv[k.idx];
}

static int g_b_ring_mask_value();         // (defined below, next to the mask)
static int pick_cfg(const ConvArgs& a, int stride, int prec = PREC_F32) {
    {
        const int v = exp_int("OSA_CONV_CFG", -1);
        if (v >= 0 && v < N_CFGS && a.CoP % g_cfgs[v].N == 0) return v;
    }
    const bool flat = (a.Ad == 1);
    const long long vox = (long long)a.B * a.Ad * a.Ah * a.Aw;
    if (flat) {
        // measured on MI355X (tools/bench_layers.py --set 2d, 1 and 2 pairs per step)
        if (stride == 2) return (a.CoP % 64 == 0) ? 14 : 12;
        // Small maps (the 1/8 and 1/16 GRU levels of the update block: 2040 / 8160 pixels, LightStereo's 24x78 level): 128-pixel
        // tiles leave most of the 256 CUs without a workgroup and every workgroup walks the whole K loop alone -- the 32- / 64-pixel
        // tiles are 2.4x faster there (gru16 256->128 @34x60: 0.061 -> 0.025 ms, gru08 384->128 @68x120: 0.094 -> 0.038 ms,
        // 768->192 1x1 @24x78: 0.045 -> 0.032 ms; tools/bench_layers.py --set gru)
        // r3 (tools/bench_gru_cfgs.sh, 4 pairs per launch, profiles/round3/gru_tile_sweep_B4.txt): the 32-pixel tile wins until the
        // 128-pixel grid has ~3 workgroups per CU -- gru08 r|z 384->256 @68x120 (510 workgroups): 0.224 -> 0.197 ms, gru08 q (255): 0.148 ->
        // 0.100 ms, gru16 r|z: 0.073 -> 0.054 ms; from 1020 workgroups on (gru04 q) the 128-pixel tile is 20 % faster
        const long long tiles128 = (vox + 127) / 128;
        if (a.CoP % 128 == 0) return (tiles128 * (a.CoP / 128) < 768) ? 11 : 9;     // 32 or 128 pixels x 128 channels
        if (a.CoP % 64 == 0) return (tiles128 * (a.CoP / 64) < 128) ? 14 : 13;       // 64 or 128 pixels x 64 channels
        return 12;                                                                   // 128 pixels x 32 channels
    }
    if (stride == 2) return (a.CoP % 128 == 0) ? 6 : ((a.CoP % 64 == 0) ? 5 : 15);
    // measured on MI355X (tools/bench_layers.py): few-tap launches (1x1x1, transposed-conv parity
    // classes) and sub-megavoxel volumes prefer the 128-voxel bricks (more workgroups in flight)
    if (a.T <= 8) return (a.CoP % 64 == 0) ? 4 : 3;
    // r4: with the weight fragments coming through the LDS ring (f16x3 / f16 modes, tile 4 in the ring mask) the 128-voxel x 64-channel tile --
    // four waves of 32 x 64 that share every fragment -- is ahead of the 256 x 64 and the 128 x 128 tiles on the stride-1 layers with 64 / 128
    // output channels (profiles/round4/tiles_128x64_ring.txt, 8 and 4 pairs: conv2 @V1 1.008-1.024 -> 0.980-0.985 ms, conv4 @V2 0.550-0.571 -> 0.522-0.526)
    if (prec != PREC_F32 && a.CoP % 64 == 0 && ((g_b_ring_mask_value() >> 4) & 1)) return 4;
    if (a.CoP % 128 == 0) return 2;                                // 128 voxels x 128 channels, 2x2 waves
    if (vox < (1ll << 20)) return (a.CoP % 64 == 0) ? 4 : 3;
    return (a.CoP % 64 == 0) ? 1 : 0;
}

// LDS image of a staged chunk: one voxel = 64 B of operands (16 fp32, or 16 hi + 16 lo fp16).
// ds_read_b128 is serviced in 16-lane groups that mix the rows of an M tile (4 rows of 8 voxels for
// TW = 8, 2 rows of 16 for TW = 16); a group is conflict free when its 16 addresses fall into 16
// distinct 16-byte slots (mod 256 B).
//  * compact (TW = 8, unit w stride): voxels 4 slots apart -> a row's 4 lanes of a group sit on slots
//    {0,4,8,12} + const, and an ODD row stride moves the 4 rows of the tile onto the 4 residues mod 4:
//    16 distinct slots with one slot of padding per row (39 KB for the 6x10x10 brick -> 4 per CU).
//  * padded (TW = 16, strided or dilated taps): voxels 5 slots apart (conflict free within a row of
//    16), row stride a multiple of 16 slots for TW = 16 and 8 (mod 16) for TW = 8.
static void finish_geometry(ConvArgs& a, int TW, bool compact) {
    if (exp_set("OSA_NOCOMPACT")) compact = false;
    int rowq;
    if (compact) {
        a.VQ = 4;
        rowq = a.LW * 4 + 1;
    } else {
        a.VQ = 5;
        const int want = (TW == 8) ? 8 : 0;              // slots mod 16
        rowq = a.LW * 5;
        while ((rowq & 15) != want) ++rowq;
        if (exp_set("OSA_NOPAD")) rowq = a.LW * 5;
    }
    a.RowQ = rowq; a.PlaneQ = a.LH * a.RowQ;
    for (int t = 0; t < a.T; ++t)
        a.toff[t] = (a.td[t] - a.dmin) * a.PlaneQ + (a.th[t] - a.hmin) * a.RowQ + (a.tw[t] - a.wmin) * a.VQ;
    a.magicW = (unsigned)((0x100000000ull + a.LW - 1) / a.LW);
    a.magicH = (unsigned)((0x100000000ull + a.LH - 1) / a.LH);
    a.magicHW = (unsigned)((0x100000000ull + (unsigned long long)a.LH * a.LW - 1) / ((unsigned long long)a.LH * a.LW));
    a.dbg = exp_int("OSA_DBG", 0);
}

static size_t brick_bytes(ConvArgs& a, const KernelCfg& k) {
    int dmax = -128, hmax = -128, wmax = -128;
    a.dmin = a.hmin = a.wmin = 127;
    for (int t = 0; t < a.T; ++t) {
        a.dmin = a.td[t] < a.dmin ? a.td[t] : a.dmin; dmax = a.td[t] > dmax ? a.td[t] : dmax;
        a.hmin = a.th[t] < a.hmin ? a.th[t] : a.hmin; hmax = a.th[t] > hmax ? a.th[t] : hmax;
        a.wmin = a.tw[t] < a.wmin ? a.tw[t] : a.wmin; wmax = a.tw[t] > wmax ? a.tw[t] : wmax;
    }
    a.LD = (k.TD - 1) * a.isd + (dmax - a.dmin) + 1;
    a.LH = (k.TH - 1) * a.ish + (hmax - a.hmin) + 1;
    a.LW = (k.TW - 1) * a.isw + (wmax - a.wmin) + 1;
    return (size_t)a.LD * a.LH * (a.LW * VS + 64) * sizeof(float);   // upper bound incl. row padding
}

#ifdef OSA_EXPERIMENTS
// conv_pipe.hip -- experiments build only (tools/build_variant.sh, OSA_PIPE=1): measured slower than the form below on every
// layer it covers (profiles/round2/pipe_ablation.txt), so the shipped library neither links nor selects it.
void (*pipe_kernel(int tile, int ring, int outs))(const ConvArgs);

// Persistent LDS-DMA pipelined launch (conv_kernel.h, PIPE = 1) for the layers it covers: f16x3, split input, unit
// stride, no gate, output channels filling one N tile of 32 / 64 / 128.  Returns 1 when launched, 0 when not eligible.
static int launch_conv_pipe(ConvArgs& a, int stride, int prec, hipStream_t st, const char* what) {
    if (prec != PREC_F16X3 || !(a.act & OSA_IN_SPLIT) || stride != 1 || a.isd != 1 || a.ish != 1 || a.isw != 1) return 0;
    if (a.gate || a.rx || a.os != 1 || a.Ci % CC != 0 || !exp_int("OSA_PIPE", 0)) return 0;
    if (a.Ad < 2) return 0;                            // flat (2-D) maps: the 3-D bricks below would idle 3 of their 4 planes
    const int tile = (a.CoP == 32) ? 0 : ((a.CoP == 64) ? 1 : ((a.CoP == 128) ? 2 : -1));
    if (tile < 0) return 0;
    static const KernelCfg shapes[3] = { {"pipe_2x1_4x1_8x8", 256, 32, 4, 8, 8, 256, 0, 3, -1},
                                         {"pipe_2x2_4x1_8x8", 256, 64, 4, 8, 8, 256, 0, 3, -1},
                                         {"pipe_2x2_2x2_8x8", 128, 128, 2, 8, 8, 256, 0, 3, -1} };
    const KernelCfg& k = shapes[tile];
    if ((long long)a.Di * a.Hi * a.Wi * a.xCs * 4 >= (1ll << 32)) return 0;          // per-row buffer descriptors: 32-bit byte counts
    a.tilesD = cdiv(a.Ad, k.TD); a.tilesH = cdiv(a.Ah, k.TH); a.tilesW = cdiv(a.Aw, k.TW);
    (void)brick_bytes(a, k);
    if (a.LW * 4 > 64 || (long long)a.LD * a.LH * a.LW >= 65536) return 0;          // one LDS-DMA instruction per brick row
    finish_geometry(a, k.TW, true);
    if (a.VQ != 4) return 0;
    const size_t brick = (size_t)a.LD * a.PlaneQ * sizeof(float4);
    const size_t epi = (size_t)(k.threads / 64) * 32 * 36 * sizeof(float);
    if (brick < epi || 2 * brick > 80 * 1024) return 0;                              // epilogue tiles live in one buffer; 2 workgroups per CU
    a.cps = 1; a.dma = 1;
    const long long nitems = (long long)a.B * a.tilesD * a.tilesH * a.tilesW;
    if (nitems >= (1ll << 31)) return 0;
    if (a.act & OSA_OUT_SPLIT) {
        OSA_REQUIRE(a.Co % 16 == 0 && a.yCs % 16 == 0 && ((size_t)a.y & 15) == 0, "%s: split output needs Co, yCs %% 16 == 0", what);
        if (a.res) OSA_REQUIRE(a.act & OSA_RES_SPLIT, "%s: a split output takes a split residual", what);
    }
    if ((a.act & OSA_RES_SPLIT) && a.res) OSA_REQUIRE(a.Co % 16 == 0 && a.rCs % 16 == 0 && ((size_t)a.res & 15) == 0,
                                                      "%s: split residual needs Co, rCs %% 16 == 0", what);
    {
        const long long ovox = (long long)a.Do * a.Ho * a.Wo;
        const int cs = a.yCs > a.rCs ? a.yCs : a.rCs;
        OSA_REQUIRE(ovox * cs < (1ll << 31), "%s: one batch item of the output exceeds 2^31 elements", what);
    }
    void (*fn)(const ConvArgs) = pipe_kernel(tile, a.T % 3 == 0, (a.act & OSA_OUT_SPLIT) != 0);
    if (!fn) return 0;
    const int lds = (int)(2 * brick);
    static bool attr_set[3][2][2];
    bool& done = attr_set[tile][a.T % 3 == 0][(a.act & OSA_OUT_SPLIT) != 0];
    if (!done) { (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done = true; }
    const int resident = exp_int("OSA_PIPE_WGS", 2) * 256;                           // 2 workgroups per CU x 256 CUs
    a.dbg = exp_int("OSA_DBG", 0);
    const unsigned grid = (unsigned)(nitems < resident ? nitems : resident);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(k.threads + 64), lds, st, a);         // + the loader wave
    OSA_LAUNCH_CHECK(what);
    return 1;
}

#endif   // OSA_EXPERIMENTS

// Start-up stagger of the first dispatch wave (conv_kernel.h).  slots = workgroups that share a CU (LDS and wave limits), period =
// what one workgroup lasts when it has the CU to itself in every phase -- modelled from the launch geometry, not measured:
//   taps   : MFMAs per wave x 32 (f16x3: 3 per product, 8 passes x 4 cycles) or x 64 (f32: 32x32x2, 16 passes) cycles
//   memory : bytes the workgroup moves (staged brick x chunks + output tile + residual) at its share of ~4 TB/s
// Slot s starts s * period / slots late.  Launches that do not fill the chip twice over gain nothing and are left alone.
static int resident_workgroups(void (*fn)(const ConvArgs), int threads, size_t lds) {
    // what the hardware will co-schedule on one CU (registers, LDS, wave slots), memoised per (kernel, LDS size)
    struct Key { const void* f; size_t l; int n; };
    static Key cache[64]; static int used = 0;
    for (int i = 0; i < used; ++i) if (cache[i].f == (const void*)fn && cache[i].l == lds) return cache[i].n;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)fn, threads, lds) != hipSuccess) { (void)hipGetLastError(); n = 1; }
    if (used < 64) cache[used++] = Key{(const void*)fn, lds, n};
    return n;
}

static void set_stagger(ConvArgs& a, const KernelCfg& k, void (*fn)(const ConvArgs), size_t lds, long long nwg, bool deconv, int prec) {
    a.stag_ticks = 0; a.stag_n = 0; a.stag_cus = 256;
    const int mode = exp_int("OSA_STAG", 0);      // measured neutral (profiles/round3/stagger_sweep_B8.txt): experiments build only
    if (!mode) return;
    const int waves = k.threads / 64;
    int slots = resident_workgroups(fn, k.threads, lds);
    { const int o = exp_int("OSA_STAG_K", 0); if (o) slots = o; }
    if (slots < 2 || nwg < (long long)2 * slots * 256) return;
    // period model (cycles at ~2 GHz -> 10 ns ticks)
    const double mfma_per_wave = (double)a.nchunks * a.T * ((double)k.M * k.N / 1024.0 / waves) * (prec == PREC_F16X3 ? 3.0 : 8.0);
    const double tap_us = mfma_per_wave * (prec == PREC_F16X3 ? 32.0 : 64.0) / 2000.0;
    const double bytes = (double)a.LD * a.LH * a.LW * 64.0 * a.nchunks + (double)k.M * (deconv ? 8 : 1) * k.N * 4.0 * (a.res || a.rx ? 2 : 1);
    const double mem_us = bytes / (4.0e6 / 256.0);         // 4 TB/s over 256 CUs = 15.6 KB/us per CU
    // de-phased steady state: the CU's matrix pipes need slots * tap_us per round of its workgroups, its memory share slots * mem_us,
    // and a single workgroup cannot finish faster than tap_us + mem_us  ->  consecutive slots start max(...) / slots apart
    double step_us = tap_us > mem_us ? tap_us : mem_us;
    if ((tap_us + mem_us) / slots > step_us) step_us = (tap_us + mem_us) / slots;
    { const int o = exp_int("OSA_STAG_PCT", 100); step_us *= o / 100.0; }
    { const int o = exp_int("OSA_STAG_US10", 0); if (o) step_us = o / 10.0; }
    a.stag_ticks = (int)(step_us * 100.0 + 0.5);
    a.stag_n = slots * 256; a.stag_cus = 256;
    if (exp_int("OSA_STAG_PRINT", 0)) fprintf(stderr, "[stagger] %s slots %d tap %.2f us mem %.2f us step %.2f us\n", k.name, slots, tap_us, mem_us, step_us);
}

// d-marching form of the 3x3x3 stride-1 32-output-channel layers (f16x3): conv_march.hip.  1 = launched, 0 = not eligible, -1 = error
int launch_conv_march(ConvArgs& a, hipStream_t st, const char* what);
// d-marching form of the 3x3x3 stride-2 64-output-channel layers (f16x3, split tensors): conv_march.hip / conv_march_s2.h.  Switch: bit 29 of
// osa_conv_b_ring_mask (A/B runs and the parity test against the brick form)
int launch_conv_march_s2(ConvArgs& a, hipStream_t st, const char* what);
void wgrad_set_multi_tile(int on);   // bit 27 of the mask: the multi-tile form of the f16x3 / f16 weight gradient (csrc/wgrad.hip wgrad_mt_kernel); 0 = single-tile kernel
void march_s2_set_waves(int w);     // bit 28 of the mask: the 4-wave 2 x 32 column (two workgroups per CU) instead of the 8-wave 4 x 32 column (one per CU)

// Which tile configurations take their B operands through the LDS ring (osa_conv_b_ring_mask; bit i = conv_cfgs.def entry i, bit 30 = the
// fused transposed convs).  Default = the tiles where the ring measured ahead at 8 AND at 4 pairs per launch (profiles/round4/
// b_ring_layers_ab.txt, b_ring_ablation_and_tiles.txt): the stride-1 tiles with 64 output channels or more per workgroup whose four waves
// share every fragment -- 1 (256 x 64: GwcNet conv2 1.067 -> 1.001 ms), 2 (128 x 128: conv4 0.619 -> 0.569), 3 / 4 (few-tap launches: redir 1x1x1
// +3-7 %), 13 (2-D 128 x 64: the 31 quarter-resolution 64 -> 64 layers 0.157 -> 0.152).  Left on the per-wave stream: the stride-2 tiles 5 / 6 /
// 14 / 15 (3-6 MFMAs per step: one barrier + one transfer per 96-192 matrix cycles costs more than the stream, conv3 -36 %), the 32-channel
// tiles 0 / 7 / 12 (first 32 -> 32: -11 %), 9 (2 x 2 waves share a fragment only pairwise: 128 -> 128 @1/4 -3 ... +1 %), 11 (no sharing at
// all), the fused transposed convs (+-1 %).
static int g_b_ring_mask = (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 13) | (1 << 29) | (1 << 27);
// (measured and not kept: tile 9 on the ring for long K loops only, >= 16 input chunks -- the GRU gate convs 384 -> 128 / 256 @1/4 gain 0 ... +5 % as
// single layers, and the IGEV x 32 loop LOSES 1.3 % with it, StereoBase 0.4 %: profiles/round4/b_ring_tile9_long_k.txt)
static long long g_b_ring_launches = 0;
static int g_b_ring_mask_value() { return g_b_ring_mask; }

static int launch_conv(ConvArgs& a, int stride, int prec, hipStream_t st, const char* what,
                       const KernelCfg* forced = nullptr) {
#ifdef OSA_EXPERIMENTS
    if (!forced) {
        const int r = launch_conv_pipe(a, stride, prec, st, what);
        if (r != 0) return r < 0 ? r : 0;
    }
#endif
    int ci = forced ? 0 : pick_cfg(a, stride, prec);
    if (!forced && brick_bytes(a, g_cfgs[ci]) > 160 * 1024) {
        // e.g. a stride-2 3x3x3 layer whose output depth collapses to 1: fall back to the small bricks
        static const int fallback[] = {5, 15, 3, 11};
        for (int f : fallback)
            if (a.CoP % g_cfgs[f].N == 0 && brick_bytes(a, g_cfgs[f]) <= 160 * 1024) { ci = f; break; }
    }
    const KernelCfg* kp = forced ? forced : &g_cfgs[ci];
    if (!forced && (ci == 11 || ci == 14) && a.Ad == 1 && !(a.act & OSA_OUT_SPLIT)) {
        // small 2-D map on the 32- / 64-pixel tiles: still few workgroups, each alone with a long K loop -> split K inside the
        // workgroup when the chunk count allows (4 groups from 16 chunks on, 2 groups from 8)
        // ... and only while the grid is still small: with >= 192 workgroups (gru16 at 4 pairs per launch: 255) the plain 32-pixel tile is
        // faster than any split (0.035 vs 0.037 / 0.045 ms for 2 / 4 groups), below ~96 workgroups four groups pay (one pair per launch)
        const int want = exp_int("OSA_KS", -1);
        // (exact-f32 mode: decided from ONE batch item's pixels, so that the summation order -- and with it every output bit -- does not depend
        // on how many pairs share the launch; tests/test_gpu_parity.py::test_gwcnet_batch_invariance_and_odd_size)
        const long long items = (prec == PREC_F32) ? 1 : a.B;
        const long long nwg = ((items * a.Ad * a.Ah * a.Aw + g_cfgs[ci].M - 1) / g_cfgs[ci].M) * (a.CoP / g_cfgs[ci].N);
        int ks = (a.nchunks % 4 == 0 && a.nchunks >= 16) ? 4 : ((a.nchunks % 2 == 0 && a.nchunks >= 8) ? 2 : 1);
        if (nwg >= 192) ks = 1;
        else if (nwg >= 96 && ks == 4) ks = 2;
        if (want >= 0) ks = (want > 1 && a.nchunks % want == 0) ? want : 1;
        if (ks == 4) kp = &g_ks_cfgs[ci == 11 ? 0 : 1];
        else if (ks == 2) kp = &g_ks_cfgs[ci == 11 ? 2 : 3];
    }
    const KernelCfg& k = *kp;
    a.tilesD = cdiv(a.Ad, k.TD); a.tilesH = cdiv(a.Ah, k.TH); a.tilesW = cdiv(a.Aw, k.TW);
    (void)brick_bytes(a, k);
    OSA_REQUIRE((long long)a.LD * a.LH * a.LW < 65536, "%s: LDS brick too large", what);
    finish_geometry(a, k.TW, k.TW == 8 && a.isw == 1);
    const size_t brick = (size_t)a.LD * a.PlaneQ * sizeof(float4);
    OSA_REQUIRE(brick <= 160 * 1024, "%s: LDS brick %dx%dx%d needs %zu B (> 160 KiB)", what, a.LD, a.LH, a.LW, brick);
    // several channel chunks per staging pass (fewer barriers, more loads in flight) while the
    // workgroup stays under ~40 KiB of LDS, i.e. as long as it does not cost residency
    a.cps = 1;
    {
        const int want = exp_int("OSA_CPS", 4);
        const size_t cap = (size_t)exp_int("OSA_CPS_LDS", 40 * 1024);
        while (a.cps < want && a.cps < a.nchunks && (size_t)(a.cps + 1) * brick <= cap) ++a.cps;
    }
    size_t lds = brick * a.cps;
    if (k.ks > 1) {                                  // one chunk per group and pass; the partial tiles meet in the same LDS afterwards
        a.cps = 1;
        const size_t red = (size_t)(k.ks - 1) * (k.threads / k.ks / 64) * ((size_t)k.M * k.N / (k.threads / k.ks / 64) / 1024) * 16 * 64 * sizeof(float);
        lds = brick * k.ks;
        if (lds < red) lds = red;
    }
#ifndef OSA_TB
#define OSA_TB 1
#endif
    const size_t epi = (size_t)(k.threads / 64) * OSA_TB * 32 * 36 * sizeof(float);   // wave-private transpose tiles of the epilogue
    if (lds < epi) lds = epi;
    { const size_t m = (size_t)exp_int("OSA_LDS_MIN", 0); if (m > lds) lds = m; }   // experiments: cap residency
    const long long nblk = (long long)a.B * a.tilesD * a.tilesH * a.tilesW;
    OSA_REQUIRE(nblk < (1ll << 31), "%s: grid too large", what);
    {   // the epilogue addresses one batch item with 32-bit element offsets
        const long long ovox = (long long)a.Do * a.Ho * a.Wo;
        const int cs = a.yCs > a.rCs ? (a.yCs > a.gCs ? a.yCs : a.gCs) : (a.rCs > a.gCs ? a.rCs : a.gCs);
        OSA_REQUIRE(ovox * cs < (1ll << 31), "%s: one batch item of the output exceeds 2^31 elements", what);
        OSA_REQUIRE((long long)a.Di * a.Hi * a.Wi * a.xCs < (1ll << 31), "%s: one batch item of the input exceeds 2^31 elements", what);
    }
    if (prec == PREC_F16) {
        // fp16 tensors (OSA_IN_F16 / OSA_OUT_F16 / OSA_RES_F16): channel strides arrive here in FLOAT units (conv3d_impl / deconv3d_impl halved them)
        OSA_REQUIRE(!a.rx, "%s: the f16 mode has no fused redir branch", what);
        if (a.act & OSA_IN_SPLIT) OSA_REQUIRE(a.Ci % 4 == 0 && a.xCs % 4 == 0, "%s: fp16 input needs channels %% 8 == 0", what);
        if (a.act & OSA_OUT_SPLIT) {
            OSA_REQUIRE(a.Co % 8 == 0 && a.yCs % 4 == 0 && !a.gate && ((size_t)a.y & 15) == 0, "%s: fp16 output needs Co, yCs %% 8 == 0 and no gate", what);
            if (a.res) OSA_REQUIRE(a.act & OSA_RES_SPLIT, "%s: an fp16 output takes an fp16 residual", what);
        }
        if ((a.act & OSA_RES_SPLIT) && a.res) OSA_REQUIRE(a.Co % 4 == 0 && a.rCs % 2 == 0 && ((size_t)a.res & 7) == 0, "%s: fp16 residual needs Co %% 4 == 0, rCs %% 4 == 0", what);
        if ((a.act & OSA_RES_SPLIT) && (a.act & OSA_OUT_SPLIT) && a.res) OSA_REQUIRE(a.rCs % 4 == 0 && ((size_t)a.res & 15) == 0, "%s: fp16 residual of an fp16 output needs rCs %% 8 == 0", what);
    } else if (a.act & (OSA_IN_SPLIT | OSA_OUT_SPLIT | OSA_RES_SPLIT | OSA_REDIR_SPLIT)) {
        OSA_REQUIRE(prec == PREC_F16X3, "%s: split activation tensors exist in the f16x3 mode only", what);
        if (a.act & OSA_IN_SPLIT) OSA_REQUIRE(a.Ci % 16 == 0, "%s: split input needs Ci %% 16 == 0 (got %d)", what, a.Ci);
        if (a.act & OSA_OUT_SPLIT) OSA_REQUIRE(a.Co % 16 == 0 && a.yCs % 16 == 0 && !a.gate && ((size_t)a.y & 15) == 0,
                                               "%s: split output needs Co, yCs %% 16 == 0 and no gate", what);
        if ((a.act & OSA_OUT_SPLIT) && a.res) OSA_REQUIRE(a.act & OSA_RES_SPLIT, "%s: a split output takes a split residual", what);
        if ((a.act & OSA_RES_SPLIT) && a.res) OSA_REQUIRE(a.Co % 16 == 0 && a.rCs % 16 == 0 && ((size_t)a.res & 15) == 0,
                                                          "%s: split residual needs Co, rCs %% 16 == 0", what);
        if ((a.act & OSA_REDIR_SPLIT) && a.rx) OSA_REQUIRE(a.rCi % 16 == 0, "%s: split redir input needs channels %% 16 == 0", what);
    }
    // LDS-DMA staging of split inputs (one global_load_lds_dwordx4 per brick row, no VGPR round trip): +1 % on the whole GwcNet step
    // (interleaved A/B, profiles/round3/ab_dma_s2u.txt); OSA_DMA=0 in the experiments build switches it off
    a.dma = (exp_int("OSA_DMA", 1) && prec != PREC_F32 && (a.act & OSA_IN_SPLIT) && a.VQ == 4 && a.LW * 4 <= 64) ? 1 : 0;
    // tap counts that are multiples of 3 (3x3x3, 3x3) run the B-ring pipeline
    const bool no_ring = exp_set("OSA_NORING");
    const KernelFns& kf = fns_of(k, prec);
    void (*fn)(const ConvArgs) = (kf.fn3 && a.T % 3 == 0 && !no_ring) ? kf.fn3 : kf.fn;
    if (a.act & OSA_OUT_SPLIT) {
        fn = (kf.fns3 && a.T % 3 == 0 && !no_ring) ? kf.fns3 : kf.fns;
        OSA_REQUIRE(fn != nullptr || kf.fnbs != nullptr, "%s: this tile configuration has no split- / fp16-output variant", what);
    }
    OSA_REQUIRE(fn != nullptr || kf.fnb != nullptr, "%s: tile configuration %s is not built for this arithmetic mode", what, k.name);
    // B operands through the LDS ring (conv_kernel.h, BL = 1; f16x3 / f16 modes): a 4-slot ring of one tap step's fragments
    // (2 KB per 32 output channels of the workgroup) above the bricks and the epilogue tiles.  Bit-identical results.
    a.ringQ = 0;
    {
        void (*fb)(const ConvArgs) = (a.act & OSA_OUT_SPLIT) ? kf.fnbs : kf.fnb;
        const size_t ring = (size_t)4 * 2 * (k.N / 32) * 1024;
        const int bit = (k.table == 0) ? (int)(&k - g_cfgs) : 30;     // osa_conv_b_ring_mask: conv_cfgs.def index, 30 = the fused transposed convs
        const bool use = fb != nullptr && k.ks <= 1 && (((g_b_ring_mask >> bit) & 1) || fn == nullptr) && lds + ring <= 160 * 1024;   // (fn == nullptr: a ring-only tile)
        if (use) { fn = fb; a.ringQ = (int)(lds / 16); lds += ring; ++g_b_ring_launches; }
        OSA_REQUIRE(fn != nullptr, "%s: tile configuration %s exists in the ring form only and its ring does not fit", what, k.name);
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)nblk, a.CoP / k.N), block(k.threads);
    set_stagger(a, k, fn, lds, (long long)grid.x * grid.y, forced != nullptr, prec);
    hipLaunchKernelGGL(fn, grid, block, lds, st, a);
    OSA_LAUNCH_CHECK(what);
    return 0;
}

// ------------------------------------------------------------------ packing --
// dst float index: ((((ch*T + t)*JO + j)*2 + h)*CoP + co)*4 + e   <-  W_t[ci = ch*16 + 8j + 4h + e][co]
struct PackArgs {
    const float* src; float* dst;
    int Ci, Co, CoP, kd, kh, kw, T, nchunks, transposed;
    signed char kz[MAX_TAPS], ky[MAX_TAPS], kx[MAX_TAPS];   // kernel index of every tap
};

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs p) {
    const size_t total = (size_t)p.nchunks * p.T * JO * 2 * p.CoP * 4;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 3; size_t r = i >> 2;
    const int co = r % p.CoP; r /= p.CoP;
    const int h = r & 1; r >>= 1;
    const int j = r % JO; r /= JO;
    const int t = r % p.T; const int ch = r / p.T;
    const int ci = ch * CC + 8 * j + 4 * h + e;
    float v = 0.f;
    if (ci < p.Ci && co < p.Co) {
        const size_t kvol = (size_t)p.kd * p.kh * p.kw;
        const size_t kidx = ((size_t)p.kz[t] * p.kh + p.ky[t]) * p.kw + p.kx[t];
        v = p.transposed ? p.src[((size_t)ci * p.Co + co) * kvol + kidx]
                         : p.src[((size_t)co * p.Ci + ci) * kvol + kidx];
    }
    p.dst[i] = v;
}

// f16x3 image of the same buffer: 16-byte unit index ((((ch*T + t)*2 + hl)*2 + kg)*CoP + co) holds the 8 fp16
// hi (hl=0) or lo (hl=1) parts of  wscale * W_t[ci = ch*16 + 8*kg + e][co], e = 0..7.
__global__ __launch_bounds__(256) void pack_weights_f16x3_kernel(const PackArgs p, float wscale) {
    const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;      // fp16 elements
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 7; size_t r = i >> 3;
    const int co = r % p.CoP; r /= p.CoP;
    const int kg = r & 1; r >>= 1;
    const int hl = r & 1; r >>= 1;
    const int t = r % p.T; const int ch = r / p.T;
    const int ci = ch * CC + 8 * kg + e;
    float v = 0.f;
    if (ci < p.Ci && co < p.Co) {
        const size_t kvol = (size_t)p.kd * p.kh * p.kw;
        const size_t kidx = ((size_t)p.kz[t] * p.kh + p.ky[t]) * p.kw + p.kx[t];
        v = wscale * (p.transposed ? p.src[((size_t)ci * p.Co + co) * kvol + kidx]
                                   : p.src[((size_t)co * p.Ci + ci) * kvol + kidx]);
    }
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16*>(p.dst)[i] = hl ? (_Float16)(v - (float)hi) : hi;
}

// f16 image (PREC_F16): chunks of 32 input channels, 16-byte unit index ((((ch*T + t)*2 + hl)*2 + kg)*CoP + co) holds the 8 fp16 values
// W_t[ci = ch*32 + 16*hl + 8*kg + e][co], e = 0..7, rounded to nearest even (what autocast's cast does to the weights); no scaling.
__global__ __launch_bounds__(256) void pack_weights_f16_kernel(const PackArgs p) {
    const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;      // fp16 elements
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 7; size_t r = i >> 3;
    const int co = r % p.CoP; r /= p.CoP;
    const int kg = r & 1; r >>= 1;
    const int hl = r & 1; r >>= 1;
    const int t = r % p.T; const int ch = r / p.T;
    const int ci = ch * 32 + 16 * hl + 8 * kg + e;
    float v = 0.f;
    if (ci < p.Ci && co < p.Co) {
        const size_t kvol = (size_t)p.kd * p.kh * p.kw;
        const size_t kidx = ((size_t)p.kz[t] * p.kh + p.ky[t]) * p.kw + p.kx[t];
        v = p.transposed ? p.src[((size_t)ci * p.Co + co) * kvol + kidx] : p.src[((size_t)co * p.Ci + ci) * kvol + kidx];
    }
    reinterpret_cast<_Float16*>(p.dst)[i] = (_Float16)v;
}

// Power-of-two weight pre-scale derived ON THE DEVICE from max |w| (training: weights change every optimizer step; a host-side scale
// would cost one synchronisation per layer, role and step and would rule out hipGraph capture of the step): largest |w| * wscale in
// [2^13, 2^14) -- floor(log2(16384 / amax)) clamped to [-14, 40], from the exponent bits (no transcendental).  amax 0 / inf / NaN: 1.
__device__ __forceinline__ float auto_wscale(float amax) {
    const unsigned b = __builtin_bit_cast(unsigned, amax);
    const int eb = (int)((b >> 23) & 0xffu);
    if (eb == 0 || eb == 255) return 1.f;
    int k = 14 - (eb - 127) - ((b & 0x7fffffu) ? 1 : 0);
    k = k < -14 ? -14 : (k > 40 ? 40 : k);
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}
__global__ __launch_bounds__(256) void pack_weights_f16x3_auto_kernel(const PackArgs p, const float* __restrict__ amax, float* __restrict__ scale_out) {
    const float wscale = auto_wscale(*amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = wscale; scale_out[1] = 1.0f / wscale; }
    const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 7; size_t r = i >> 3;
    const int co = r % p.CoP; r /= p.CoP;
    const int kg = r & 1; r >>= 1;
    const int hl = r & 1; r >>= 1;
    const int t = r % p.T; const int ch = r / p.T;
    const int ci = ch * CC + 8 * kg + e;
    float v = 0.f;
    if (ci < p.Ci && co < p.Co) {
        const size_t kvol = (size_t)p.kd * p.kh * p.kw;
        const size_t kidx = ((size_t)p.kz[t] * p.kh + p.ky[t]) * p.kw + p.kx[t];
        v = wscale * (p.transposed ? p.src[((size_t)ci * p.Co + co) * kvol + kidx]
                                   : p.src[((size_t)co * p.Ci + ci) * kvol + kidx]);
    }
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16*>(p.dst)[i] = hl ? (_Float16)(v - (float)hi) : hi;
}

static inline int pad32(int c) { return (c + 31) / 32 * 32; }
static inline int nchunks_of(int ci) { return (ci + CC - 1) / CC; }
static inline size_t packed_floats(int Ci, int Co, int T) {
    return (size_t)nchunks_of(Ci) * T * JO * 2 * pad32(Co) * 4;
}
static inline size_t slack_floats(int Co) { return (size_t)4 * JO * 2 * pad32(Co) * 4; }   // up to 4 prefetched taps

// transposed-conv parity class: taps of one dimension. o = 2a+par ; i = a + delta ; kernel index kk
static int deconv_dim_taps(int k, int pad, int par, int* delta, int* kk) {
    int n = 0;
    for (int t = 0; t < k; ++t) {
        const int num = par + pad - t;
        if (((num % 2) + 2) % 2 != 0) continue;
        delta[n] = (num >= 0) ? num / 2 : -((-num) / 2);
        kk[n] = t;
        ++n;
    }
    return n;
}

// ------------------------------------------------------------------ small Co --
// Classifier heads (32 -> 1): N is far too small for the matrix cores, so this is a VALU kernel
// on the same LDS brick: one thread per output voxel, all taps x 16-channel chunks read from LDS
// as float4, the (tiny) weight set re-ordered into LDS as [chunk][tap][co][16] and read by
// broadcast.  HBM traffic = one pass over the input; LDS-read bound.
// WG = false: weights in the reference layout, re-ordered into LDS by every workgroup.
// WG = true : weights pre-packed [chunk][tap][co][16] in global memory; the address is wave-uniform, so
//             they arrive through the scalar cache (s_load_dwordx16) and feed the FMAs as SGPR operands --
//             half the LDS instructions per FMA of the WG = false form.
template <int CO, bool WG>
__global__ __launch_bounds__(256) void conv_small_co_tiled_kernel(const ConvArgs p, const float* __restrict__ wref,
                                                                  const float* __restrict__ bias) {
    constexpr int TD = 4, TH = 8, TW = 8;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    float4* wl4 = smem + (size_t)p.LD * p.PlaneQ;             // [nchunks][T][CO][16 floats]
    float* wl = reinterpret_cast<float*>(wl4);

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int twi = bid % p.tilesW; bid /= p.tilesW;
    const int thi = bid % p.tilesH; bid /= p.tilesH;
    const int tdi = bid % p.tilesD;
    const int b = bid / p.tilesD;
    const int a0d = tdi * TD, a0h = thi * TH, a0w = twi * TW;
    const int g0d = a0d + p.dmin, g0h = a0h + p.hmin, g0w = a0w + p.wmin;

    const int nw = WG ? 0 : p.nchunks * p.T * CO * 16;
    for (int i = tid; i < nw; i += 256) {
        const int e = i & 15; int r = i >> 4;
        const int co = r % CO; r /= CO;
        const int t = r % p.T; const int ch = r / p.T;
        const int ci = ch * CC + e;
        wl[i] = (ci < p.Ci) ? wref[((size_t)co * p.Ci + ci) * p.T + t] : 0.f;
    }
    const int tw_ = tid % TW, th_ = (tid / TW) % TH, td_ = tid / (TW * TH);
    const int abase = td_ * p.PlaneQ + th_ * p.RowQ + tw_ * p.VQ;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = bias ? bias[o] : 0.f;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        if (ch) __syncthreads();
        stage_brick<256, PREC_F32, 1>(p, smem, 0, b, ch * CC, g0d, g0h, g0w, tid);
        __syncthreads();
        for (int t = 0; t < p.T; ++t) {
            const float4* xp = smem + abase + p.toff[t];
            const float4* wq = WG ? reinterpret_cast<const float4*>(wref) + (size_t)(ch * p.T + t) * CO * 4
                                  : wl4 + (size_t)(ch * p.T + t) * CO * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 xv = xp[q];
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const float4 wv = wq[o * 4 + q];
                    acc[o] = fmaf(xv.x, wv.x, acc[o]); acc[o] = fmaf(xv.y, wv.y, acc[o]);
                    acc[o] = fmaf(xv.z, wv.z, acc[o]); acc[o] = fmaf(xv.w, wv.w, acc[o]);
                }
            }
        }
    }
    const int ad = a0d + td_, ah = a0h + th_, aw = a0w + tw_;
    if (ad < p.Ad && ah < p.Ah && aw < p.Aw) {
        const size_t vox = (((size_t)b * p.Do + ad) * p.Ho + ah) * p.Wo + aw;
#pragma unroll
        for (int o = 0; o < CO; ++o)
            p.y[vox * p.yCs + o] = acc[o] + (p.res ? p.res[vox * p.yCs + o] : 0.f);
    }
}


// ------------------------------------------------------------------ classifier 32 -> 1, d-marching form --
// The 3x3x3 "same" convolution 32 -> 1 at the end of every aggregation head (gwcnet_disp_processor.py:76-80, psmnet_cost_processor.py:
// classif1-3).  The brick form above re-reads its 6x10x10 halo brick per 4x8x8 tile (2.34x the input; 2.7x measured) and is HBM-bound on
// those re-reads.  Here a workgroup owns a 16 x 16 pixel column and WALKS along d: every input plane (18 x 18 pixels x all 32 channels,
// 128-byte rows of full cache lines) is staged ONCE, read from LDS once per (tap, channel quad) and feeds three running sums -- the
// outputs at d-1, d, d+1, whose kd = 2, 1, 0 taps it is.  HBM traffic = (18 x 18) / (16 x 16) x (dseg + 2) / dseg = 1.27-1.37x of one
// pass; 12 FMAs per ds_read_b128 instead of 4; the next plane's loads are in flight (registers) while the current one is consumed.
// LDS image: pixel stride 9 slots of 16 B (odd: the 16 lanes of a ds_read_b128 group fall on 16 distinct slots mod 16), row stride 176
// slots (a multiple of 16, so the two pixel rows a group straddles keep that property): 18 x 176 x 16 B = 50.7 KB, 3 workgroups per CU.
// Weights: the packed [chunk][tap][co = 1][16] array, wave-uniform addresses -> scalar loads, SGPR operands.
constexpr int CM_TH = 16, CM_TW = 16, CM_LH = 18, CM_LW = 18, CM_PXQ = 9, CM_ROWQ = 176, CM_ITEMS = CM_LH * CM_LW * 8, CM_PER = (CM_ITEMS + 255) / 256;
__global__ __launch_bounds__(256) void classifier_march_kernel(const ConvArgs p, const float* __restrict__ wpk, const float* __restrict__ bias,
                                                               int dseg, int nseg) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int twi = bid % p.tilesW; bid /= p.tilesW;
    const int thi = bid % p.tilesH; bid /= p.tilesH;
    const int seg = bid % nseg;
    const int b = bid / nseg;
    const int d0 = seg * dseg, d1 = (d0 + dseg < p.Di) ? d0 + dseg : p.Di;
    const int h0 = thi * CM_TH, w0 = twi * CM_TW;
    const size_t plane = (size_t)p.Hi * p.Wi * p.xCs;
    const float* xb = p.x + (size_t)b * p.Di * plane;

    // staging items of this thread: (pixel of the 18 x 18 halo tile, channel quad) -> offset inside a plane (or -1: outside the image)
    int goff[CM_PER], loff[CM_PER];
#pragma unroll
    for (int j = 0; j < CM_PER; ++j) {
        const int i = tid + 256 * j;
        const int px = i >> 3, q = i & 7;
        const int lh = px / CM_LW, lw = px - lh * CM_LW;
        const int gh = h0 - 1 + lh, gw = w0 - 1 + lw;
        const bool in = i < CM_ITEMS && gh >= 0 && gh < p.Hi && gw >= 0 && gw < p.Wi;
        goff[j] = in ? (gh * p.Wi + gw) * p.xCs + q * 4 : -1;
        loff[j] = (i < CM_ITEMS) ? lh * CM_ROWQ + lw * CM_PXQ + q : -1;
    }
    float4 pre[CM_PER];
    auto issue = [&](int pd) {
        const float* xp = xb + (size_t)pd * plane;
#pragma unroll
        for (int j = 0; j < CM_PER; ++j)
            pre[j] = (goff[j] >= 0) ? *reinterpret_cast<const float4*>(xp + goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int tw_ = tid % CM_TW, th_ = tid / CM_TW;
    const float4* xt = smem + th_ * CM_ROWQ + tw_ * CM_PXQ;
    const float4* w4 = reinterpret_cast<const float4*>(wpk);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;          // running sums of the outputs at pd - 1, pd, pd + 1
    issue(d0 > 0 ? d0 - 1 : 0);
    for (int pd = d0 - 1; pd <= d1; ++pd) {
        if (pd >= 0 && pd < p.Di) {               // uniform over the workgroup
            __syncthreads();                      // the previous plane's readers are done
#pragma unroll
            for (int j = 0; j < CM_PER; ++j)
                if (loff[j] >= 0) smem[loff[j]] = pre[j];
            __syncthreads();
            if (pd + 1 <= d1 && pd + 1 < p.Di) issue(pd + 1);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 xv = xt[kh * CM_ROWQ + kw * CM_PXQ + q];
                        const int wi = ((q >> 2) * 27 + kh * 3 + kw) * 4 + (q & 3);        // float4 index of (chunk, tap kd = 0, quad)
                        const float4 wk0 = w4[wi], wk1 = w4[wi + 9 * 4], wk2 = w4[wi + 18 * 4];
                        a0 = fmaf(xv.x, wk2.x, a0); a0 = fmaf(xv.y, wk2.y, a0); a0 = fmaf(xv.z, wk2.z, a0); a0 = fmaf(xv.w, wk2.w, a0);
                        a1 = fmaf(xv.x, wk1.x, a1); a1 = fmaf(xv.y, wk1.y, a1); a1 = fmaf(xv.z, wk1.z, a1); a1 = fmaf(xv.w, wk1.w, a1);
                        a2 = fmaf(xv.x, wk0.x, a2); a2 = fmaf(xv.y, wk0.y, a2); a2 = fmaf(xv.z, wk0.z, a2); a2 = fmaf(xv.w, wk0.w, a2);
                    }
        }
        const int od = pd - 1, oh = h0 + th_, ow = w0 + tw_;
        if (od >= d0 && od < d1 && oh < p.Hi && ow < p.Wi) {
            const size_t vox = (((size_t)b * p.Do + od) * p.Ho + oh) * p.Wo + ow;
            p.y[vox * p.yCs] = a0 + (bias ? bias[0] : 0.f) + (p.res ? p.res[vox * p.yCs] : 0.f);
        }
        a0 = a1; a1 = a2; a2 = 0.f;
    }
}

}  // namespace osa

using namespace osa;

// ------------------------------------------------------------------ C ABI -----
extern "C" size_t osa_conv3d_packed_floats(int Ci, int Co, int kd, int kh, int kw) {
    return packed_floats(Ci, Co, kd * kh * kw) + slack_floats(Co);
}

static void launch_pack(const PackArgs& p, int prec, float wscale, hipStream_t st, const float* amax_dev = nullptr, float* scale_out = nullptr) {
    if (amax_dev) {                                    // f16x3 with the device-side scale
        const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;
        hipLaunchKernelGGL(pack_weights_f16x3_auto_kernel, dim3(total ? cdiv((long long)total, 256) : 1), dim3(256), 0, st, p, amax_dev, scale_out);
        return;
    }
    if (prec == PREC_F32) {
        const size_t total = (size_t)p.nchunks * p.T * JO * 2 * p.CoP * 4;
        if (total) hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, st, p);
    } else if (prec == PREC_F16) {
        const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;
        if (total) hipLaunchKernelGGL(pack_weights_f16_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, st, p);
    } else {
        const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;
        if (total) hipLaunchKernelGGL(pack_weights_f16x3_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, st, p, wscale);
    }
}

static int conv3d_pack_impl(const float* w_ref, float* w_packed, int Ci, int Co,
                            int kd, int kh, int kw, int prec, float wscale, void* stream,
                            int src_transposed = 0, int flip = 0, const float* amax_dev = nullptr, float* scale_out = nullptr) {
    OSA_REQUIRE(w_ref && w_packed, "conv3d_pack: NULL pointer");
    const int T = kd * kh * kw;
    OSA_REQUIRE(T >= 1 && T <= MAX_TAPS, "conv3d_pack: %dx%dx%d kernel has %d taps (max %d)", kd, kh, kw, T, MAX_TAPS);
    OSA_REQUIRE(Ci > 0 && Co > 0, "conv3d_pack: bad channels %d->%d", Ci, Co);
    PackArgs p;
    p.src = w_ref; p.dst = w_packed; p.Ci = Ci; p.Co = Co; p.CoP = pad32(Co);
    p.kd = kd; p.kh = kh; p.kw = kw; p.T = T; p.nchunks = (prec == PREC_F16) ? cdiv(Ci, 32) : nchunks_of(Ci); p.transposed = src_transposed ? 1 : 0;
    int t = 0;
    for (int z = 0; z < kd; ++z) for (int y = 0; y < kh; ++y) for (int x = 0; x < kw; ++x, ++t) {
        p.kz[t] = (signed char)(flip ? kd - 1 - z : z); p.ky[t] = (signed char)(flip ? kh - 1 - y : y);
        p.kx[t] = (signed char)(flip ? kw - 1 - x : x);
    }
    launch_pack(p, prec, wscale, (hipStream_t)stream, amax_dev, scale_out);
    OSA_LAUNCH_CHECK("conv3d_pack");
    return 0;
}

extern "C" int osa_conv3d_pack_ex_auto(const float* w_ref, float* w_packed, int Ci, int Co, int kd, int kh, int kw,
                                       int src_transposed, int flip, const float* w_amax, float* scale_out, void* stream) {
    OSA_REQUIRE(w_amax && scale_out, "conv3d_pack_ex_auto: NULL scale pointers");
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, PREC_F16X3, 1.f, stream, src_transposed, flip, w_amax, scale_out);
}

extern "C" int osa_conv3d_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                   int kd, int kh, int kw, void* stream) {
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, PREC_F32, 1.f, stream);
}

// Generic packing for backward passes: Ci/Co are the roles of the convolution that will be EXECUTED;
// src_transposed=1 reads w_ref as [Ci][Co][k] instead of [Co][Ci][k]; flip=1 mirrors the taps.
//   data-gradient of a stride-1 conv (weight [Co][Ci][k]) = conv with roles swapped:
//       pack_ex(Ci' = Co, Co' = Ci, src_transposed = 1, flip = 1), padding' = dil*(k-1) - pad
//   data-gradient of a ConvTranspose3d (weight [Ci][Co][k]) = strided conv:
//       pack_ex(Ci' = Co, Co' = Ci, src_transposed = 0, flip = 0)
extern "C" int osa_conv3d_pack_ex(const float* w_ref, float* w_packed, int Ci, int Co,
                                  int kd, int kh, int kw, int src_transposed, int flip,
                                  int f16x3, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "conv3d_pack_ex: wscale must be a positive power of two");
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, (f16x3 == 2) ? PREC_F16 : (f16x3 ? PREC_F16X3 : PREC_F32), wscale, stream,
                            src_transposed, flip);
}

extern "C" int osa_conv3d_pack_f16x3(const float* w_ref, float* w_packed, int Ci, int Co,
                                     int kd, int kh, int kw, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "conv3d_pack_f16x3: wscale must be a positive power of two");
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, PREC_F16X3, wscale, stream);
}

extern "C" int osa_conv3d_pack_f16(const float* w_ref, float* w_packed, int Ci, int Co, int kd, int kh, int kw, void* stream) {
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, PREC_F16, 1.f, stream);
}

extern "C" size_t osa_deconv3d_packed_floats(int Ci, int Co, int k) {
    // every kernel tap belongs to exactly one parity class -> k^3 taps in total
    return packed_floats(Ci, Co, k * k * k) + slack_floats(Co);
}

// class-major tap list of a stride-2 transposed conv: for class c = (pd,ph,pw) all (kz,ky,kx) that hit
// real inputs, with their input offsets delta.  Returns the total tap count (k^3).
struct DeconvTaps {
    int T; int cls_end[8];
    signed char kz[MAX_TAPS], ky[MAX_TAPS], kx[MAX_TAPS], dz[MAX_TAPS], dy[MAX_TAPS], dx[MAX_TAPS];
};
static void deconv_taps(int k, int pad, DeconvTaps& d, bool flat = false) {
    int t = 0;
    for (int cls = 0; cls < 8; ++cls) {
        int dd[4], kd_[4], dh[4], kh_[4], dw[4], kw_[4];
        // flat: a 1 x k x k kernel on a D = 1 tensor -- only the 4 classes with even d parity exist
        int nd = flat ? (((cls >> 2) & 1) ? 0 : 1) : deconv_dim_taps(k, pad, (cls >> 2) & 1, dd, kd_);
        if (flat) { dd[0] = 0; kd_[0] = 0; }
        const int nh = deconv_dim_taps(k, pad, (cls >> 1) & 1, dh, kh_);
        const int nw = deconv_dim_taps(k, pad, cls & 1, dw, kw_);
        for (int a = 0; a < nd; ++a) for (int b = 0; b < nh; ++b) for (int c = 0; c < nw; ++c, ++t) {
            d.kz[t] = (signed char)kd_[a]; d.ky[t] = (signed char)kh_[b]; d.kx[t] = (signed char)kw_[c];
            d.dz[t] = (signed char)dd[a]; d.dy[t] = (signed char)dh[b]; d.dx[t] = (signed char)dw[c];
        }
        d.cls_end[cls] = t;
    }
    d.T = t;
}

static int deconv3d_pack_impl(const float* w_ref, float* w_packed, int Ci, int Co,
                              int k, int pad, int prec, float wscale, void* stream, bool flat = false,
                              const float* amax_dev = nullptr, float* scale_out = nullptr) {
    OSA_REQUIRE(w_ref && w_packed, "deconv3d_pack: NULL pointer");
    OSA_REQUIRE(k == 3 || k == 4, "deconv3d_pack: kernel %d unsupported (3 or 4)", k);
    DeconvTaps d;
    deconv_taps(k, pad, d, flat);
    PackArgs p;
    p.src = w_ref; p.dst = w_packed; p.Ci = Ci; p.Co = Co; p.CoP = pad32(Co);
    p.kd = flat ? 1 : k; p.kh = k; p.kw = k; p.T = d.T; p.nchunks = (prec == PREC_F16) ? cdiv(Ci, 32) : nchunks_of(Ci); p.transposed = 1;
    for (int t = 0; t < d.T; ++t) { p.kz[t] = d.kz[t]; p.ky[t] = d.ky[t]; p.kx[t] = d.kx[t]; }
    launch_pack(p, prec, wscale, (hipStream_t)stream, amax_dev, scale_out);
    OSA_LAUNCH_CHECK("deconv3d_pack");
    return 0;
}

extern "C" int osa_deconv3d_pack_f16x3_auto(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad,
                                            const float* w_amax, float* scale_out, void* stream) {
    OSA_REQUIRE(w_amax && scale_out, "deconv3d_pack_f16x3_auto: NULL scale pointers");
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16X3, 1.f, stream, false, w_amax, scale_out);
}

extern "C" int osa_deconv2d_pack_f16x3_auto(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad,
                                            const float* w_amax, float* scale_out, void* stream) {
    OSA_REQUIRE(w_amax && scale_out, "deconv2d_pack_f16x3_auto: NULL scale pointers");
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16X3, 1.f, stream, true, w_amax, scale_out);
}

extern "C" int osa_deconv3d_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                     int k, int pad, void* stream) {
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F32, 1.f, stream);
}

extern "C" int osa_deconv3d_pack_f16x3(const float* w_ref, float* w_packed, int Ci, int Co,
                                       int k, int pad, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "deconv3d_pack_f16x3: wscale must be a positive power of two");
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16X3, wscale, stream);
}

extern "C" int osa_deconv3d_pack_f16(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad, void* stream) {
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16, 1.f, stream);
}
extern "C" int osa_deconv2d_pack_f16(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad, void* stream) {
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16, 1.f, stream, true);
}

// ---- 2-D transposed conv (nn.ConvTranspose2d, stride 2): the D = 1 case, 4 parity classes
extern "C" size_t osa_deconv2d_packed_floats(int Ci, int Co, int k) {
    return packed_floats(Ci, Co, k * k) + slack_floats(Co);
}

extern "C" int osa_deconv2d_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                     int k, int pad, void* stream) {
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F32, 1.f, stream, true);
}

extern "C" int osa_deconv2d_pack_f16x3(const float* w_ref, float* w_packed, int Ci, int Co,
                                       int k, int pad, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "deconv2d_pack_f16x3: wscale must be a positive power of two");
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16X3, wscale, stream, true);
}

static void set_ranges(ConvArgs& a, const osa_f16x3_ranges* r) {
    if (!r) return;
    a.in_meta = r->x_meta; a.res_meta = r->residual_meta; a.rx_meta = r->redir_meta; a.out_meta = r->y_meta;
    a.coef = r->bound_coef; a.rcoef = r->redir_bound_coef;
    a.wscale_dev = r->weight_scale;
}

// f16x3 split tensors cannot be decoded without their range blocks (scale of a split input / residual / redir input; bound
// coefficients + input range for a split output).  Checked once, in front of BOTH kernel forms (brick and d-marching).
static int check_split_ranges(const ConvArgs& a, int prec, const char* what) {
    if (prec != PREC_F16X3) return 0;
    if (a.act & OSA_IN_SPLIT) OSA_REQUIRE(a.in_meta, "%s: a split input needs its range block (osa_f16x3_ranges.x_meta)", what);
    if ((a.act & OSA_RES_SPLIT) && a.res) OSA_REQUIRE(a.res_meta, "%s: a split residual needs its range block (residual_meta)", what);
    if ((a.act & OSA_REDIR_SPLIT) && a.rx) OSA_REQUIRE(a.rx_meta, "%s: a split redir input needs its range block (redir_meta)", what);
    if (a.act & OSA_OUT_SPLIT) OSA_REQUIRE(a.out_meta && a.coef && a.in_meta,
                                           "%s: a split output needs y_meta, bound_coef and x_meta (its scale is derived from them)", what);
    return 0;
}

static int check_common(const char* what, const float* x, const float* w, float* y,
                        int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs,
                        const float* residual) {
    OSA_REQUIRE(x && w && y, "%s: NULL pointer", what);
    OSA_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0, "%s: bad dims", what);
    OSA_REQUIRE(Ci > 0 && Co > 0, "%s: bad channels %d->%d", what, Ci, Co);
    OSA_REQUIRE(Ci % 4 == 0 && xCs % 4 == 0 && xCs >= Ci, "%s: Ci=%d / xCs=%d must be multiples of 4, xCs>=Ci", what, Ci, xCs);
    OSA_REQUIRE(((size_t)x & 15) == 0, "%s: x not 16-byte aligned", what);
    OSA_REQUIRE(yCs >= Co, "%s: yCs=%d < Co=%d", what, yCs, Co);
    if (residual) OSA_REQUIRE(rCs >= Co, "%s: rCs=%d < Co=%d", what, rCs, Co);
    return 0;
}

// f16 mode: the C ABI gives channel counts / strides in ELEMENTS of each tensor; the kernel addresses every tensor through float
// pointers, so an fp16 tensor's stride (and, for the input, its channel count) is halved here.  A chunk is 32 input channels.
static int f16_units(ConvArgs& a, const char* what) {
    a.nchunks = cdiv(a.Ci, 32);
    if (a.act & OSA_IN_SPLIT) {
        OSA_REQUIRE(a.Ci % 8 == 0 && a.xCs % 8 == 0, "%s: fp16 input needs Ci, xCs %% 8 == 0 (got %d / %d)", what, a.Ci, a.xCs);
        a.Ci /= 2; a.xCs /= 2;
    }
    if (a.act & OSA_OUT_SPLIT) {
        OSA_REQUIRE(a.yCs % 8 == 0, "%s: fp16 output needs yCs %% 8 == 0 (got %d)", what, a.yCs);
        a.yCs /= 2;
    }
    if ((a.act & OSA_RES_SPLIT) && a.res) {
        OSA_REQUIRE(a.rCs % 4 == 0, "%s: fp16 residual needs rCs %% 4 == 0 (got %d)", what, a.rCs);
        a.rCs /= 2;
    }
    return 0;
}

static int conv3d_impl(const float* x, const float* w_packed,
                       const float* scale, const float* shift, const float* residual,
                       float* y,
                       int B, int Di, int Hi, int Wi, int Ci, int xCs,
                       int Co, int yCs, int rCs,
                       int kd, int kh, int kw, int stride,
                       int pad_d, int pad_h, int pad_w,
                       int dil_d, int dil_h, int dil_w,
                       const float* gate_logits, int gCs,
                       int act, float slope, int prec, float oscale, void* stream,
                       const osa_f16x3_ranges* rng = nullptr) {
    if (gate_logits) OSA_REQUIRE(gCs >= Co, "conv3d: gate stride %d < Co %d", gCs, Co);
    if (check_common("conv3d", x, w_packed, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, residual)) return -1;
    const int T = kd * kh * kw;
    OSA_REQUIRE(T >= 1 && T <= MAX_TAPS, "conv3d: %dx%dx%d kernel unsupported", kd, kh, kw);
    OSA_REQUIRE(stride == 1 || stride == 2, "conv3d: stride %d unsupported", stride);
    OSA_REQUIRE(dil_d >= 1 && dil_h >= 1 && dil_w >= 1, "conv3d: bad dilation");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = reinterpret_cast<const float4*>(w_packed); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.gate = gate_logits; a.gCs = gCs;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.xCs = xCs;
    a.isd = (Di == 1 && kd == 1) ? 1 : stride; a.ish = stride; a.isw = stride;
    a.Do = (Di + 2 * pad_d - dil_d * (kd - 1) - 1) / a.isd + 1;
    a.Ho = (Hi + 2 * pad_h - dil_h * (kh - 1) - 1) / a.ish + 1;
    a.Wo = (Wi + 2 * pad_w - dil_w * (kw - 1) - 1) / a.isw + 1;
    OSA_REQUIRE(a.Do > 0 && a.Ho > 0 && a.Wo > 0, "conv3d: empty output");
    a.Co = Co; a.yCs = yCs; a.rCs = rCs;
    a.Ad = a.Do; a.Ah = a.Ho; a.Aw = a.Wo;
    a.os = 1; a.ood = a.ooh = a.oow = 0;
    a.T = T;
    int t = 0;
    for (int z = 0; z < kd; ++z) for (int yy = 0; yy < kh; ++yy) for (int xx = 0; xx < kw; ++xx, ++t) {
        a.td[t] = (signed char)(z * dil_d - pad_d);
        a.th[t] = (signed char)(yy * dil_h - pad_h);
        a.tw[t] = (signed char)(xx * dil_w - pad_w);
    }
    a.nchunks = nchunks_of(Ci); a.CoP = pad32(Co);
    a.act = act; a.slope = slope; a.oscale = oscale;
    if (prec == PREC_F16 && f16_units(a, "conv3d")) return -1;
    set_ranges(a, rng);
    if (check_split_ranges(a, prec, "conv3d")) return -1;
    if (prec == PREC_F16X3 && kd == 3 && kh == 3 && kw == 3 && stride == 1 && a.isd == 1 && pad_d == 1 && pad_h == 1 && pad_w == 1 &&
        dil_d == 1 && dil_h == 1 && dil_w == 1) {
        const int r = launch_conv_march(a, (hipStream_t)stream, "conv3d (march)");
        if (r != 0) return r < 0 ? r : 0;
    }
    if (prec == PREC_F16X3 && kd == 3 && kh == 3 && kw == 3 && stride == 2 && a.isd == 2 && pad_d == 1 && pad_h == 1 && pad_w == 1 &&
        dil_d == 1 && dil_h == 1 && dil_w == 1 && ((g_b_ring_mask_value() >> 29) & 1)) {
        const int r = launch_conv_march_s2(a, (hipStream_t)stream, "conv3d (march, stride 2)");
        if (r != 0) return r < 0 ? r : 0;
    }
    return launch_conv(a, stride, prec, (hipStream_t)stream, "conv3d");
}

#define OSA_CONV_PARAMS                                                                         \
    const float* x, const float* w_packed, const float* scale, const float* shift,             \
    const float* residual, float* y, int B, int Di, int Hi, int Wi, int Ci, int xCs,           \
    int Co, int yCs, int rCs, int kd, int kh, int kw, int stride, int pad_d, int pad_h,        \
    int pad_w, int dil_d, int dil_h, int dil_w, const float* gate_logits, int gCs, int act, float slope
#define OSA_CONV_ARGS                                                                           \
    x, w_packed, scale, shift, residual, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, kd, kh, kw,   \
    stride, pad_d, pad_h, pad_w, dil_d, dil_h, dil_w, gate_logits, gCs, act, slope

extern "C" int osa_conv3d_ndhwc_f32(OSA_CONV_PARAMS, void* stream) {
    return conv3d_impl(OSA_CONV_ARGS, PREC_F32, 1.f, stream);
}

extern "C" int osa_conv3d_ndhwc_f16x3(OSA_CONV_PARAMS, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return conv3d_impl(OSA_CONV_ARGS, PREC_F16X3, out_scale, stream, ranges);
}

// f16 mode (PREC_F16): x / residual / y are fp32 tensors, or fp16 tensors where OSA_IN_F16 / OSA_RES_F16 / OSA_OUT_F16 say so
extern "C" int osa_conv3d_ndhwc_f16(const void* x_, const float* w_packed, const float* scale, const float* shift, const void* residual_, void* y_,
                                    int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs, int kd, int kh, int kw, int stride,
                                    int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w, const float* gate_logits, int gCs,
                                    int act, float slope, void* stream) {
    const float* x = static_cast<const float*>(x_); const float* residual = static_cast<const float*>(residual_); float* y = static_cast<float*>(y_);
    return conv3d_impl(OSA_CONV_ARGS, PREC_F16, 1.f, stream);
}

static int deconv3d_impl(const float* x, const float* w_packed,
                         const float* scale, const float* shift, const float* residual,
                         float* y,
                         int B, int Di, int Hi, int Wi, int Ci, int xCs,
                         int Co, int yCs, int rCs,
                         int k, int pad, int opad,
                         const float* gate_logits, int gCs,
                         int act, float slope, int prec, float oscale, void* stream, bool flat = false,
                         const float* rx = nullptr, int rxCs = 0, int rCi = 0, const float* rw_packed = nullptr,
                         const float* rscale = nullptr, const float* rshift = nullptr, float roscale = 1.f,
                         const osa_f16x3_ranges* rng = nullptr) {
    if (gate_logits) OSA_REQUIRE(gCs >= Co, "deconv3d: gate stride %d < Co %d", gCs, Co);
    if (check_common("deconv3d", x, w_packed, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, residual)) return -1;
    if (flat) OSA_REQUIRE(Di == 1, "deconv2d: the tensor must have D == 1 (got %d)", Di);
    OSA_REQUIRE((k == 3 && pad == 1 && opad == 1) || (k == 4 && pad == 1 && opad == 0),
                "deconv3d: only (k=3,p=1,op=1) and (k=4,p=1,op=0) with stride 2 are supported (got k=%d p=%d op=%d)", k, pad, opad);
    DeconvTaps d;
    deconv_taps(k, pad, d, flat);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = reinterpret_cast<const float4*>(w_packed);
    a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.gate = gate_logits; a.gCs = gCs;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.xCs = xCs;
    a.Do = flat ? 1 : (Di - 1) * 2 - 2 * pad + k + opad; a.Ho = (Hi - 1) * 2 - 2 * pad + k + opad; a.Wo = (Wi - 1) * 2 - 2 * pad + k + opad;
    a.Co = Co; a.yCs = yCs; a.rCs = rCs;
    a.Ad = (a.Do + 1) / 2; a.Ah = (a.Ho + 1) / 2; a.Aw = (a.Wo + 1) / 2;     // a-space covers every output parity
    a.isd = a.ish = a.isw = 1;
    a.os = 2;
    a.T = d.T;
    for (int t = 0; t < d.T; ++t) { a.td[t] = d.dz[t]; a.th[t] = d.dy[t]; a.tw[t] = d.dx[t]; }
    for (int c = 0; c < 8; ++c) a.cls_end[c] = d.cls_end[c];
    a.nchunks = nchunks_of(Ci); a.CoP = pad32(Co);
    a.act = act; a.slope = slope; a.oscale = oscale;
    if (prec == PREC_F16 && f16_units(a, flat ? "deconv2d" : "deconv3d")) return -1;
    if (rx) {
        OSA_REQUIRE(!flat && !residual && !gate_logits, "deconv3d_redir: residual / gate cannot be combined with the fused redir branch");
        OSA_REQUIRE(rw_packed && rCi > 0 && rCi <= 64 && rCi % 4 == 0 && rxCs >= rCi && rxCs % 4 == 0 && ((size_t)rx & 15) == 0,
                    "deconv3d_redir: redir input needs <= 64 channels (multiple of 4), stride >= channels, 16-byte alignment (Ci=%d stride=%d)", rCi, rxCs);
        OSA_REQUIRE((long long)a.Do * a.Ho * a.Wo * rxCs < (1ll << 31), "deconv3d_redir: redir input too large");
        a.rx = rx; a.rxCs = rxCs; a.rCi = rCi; a.rw = reinterpret_cast<const float4*>(rw_packed);
        a.rscale = rscale; a.rshift = rshift; a.roscale = roscale;
    }
    set_ranges(a, rng);
    if (check_split_ranges(a, prec, flat ? "deconv2d" : "deconv3d")) return -1;
    return launch_conv(a, 1, prec, (hipStream_t)stream, flat ? "deconv2d" : "deconv3d",
                       flat ? &g_deconv_flat_cfg : (rx ? (rCi > 32 ? &g_deconv_redir64_cfg : &g_deconv_redir_cfg) : &g_deconv_cfg));
}

#define OSA_DECONV_PARAMS                                                                       \
    const float* x, const float* w_packed, const float* scale, const float* shift,             \
    const float* residual, float* y, int B, int Di, int Hi, int Wi, int Ci, int xCs,           \
    int Co, int yCs, int rCs, int k, int pad, int opad, const float* gate_logits, int gCs, int act, float slope
#define OSA_DECONV_ARGS                                                                         \
    x, w_packed, scale, shift, residual, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, k, pad, opad, \
    gate_logits, gCs, act, slope

extern "C" int osa_deconv3d_ndhwc_f32(OSA_DECONV_PARAMS, void* stream) {
    return deconv3d_impl(OSA_DECONV_ARGS, PREC_F32, 1.f, stream);
}

extern "C" int osa_deconv3d_ndhwc_f16x3(OSA_DECONV_PARAMS, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return deconv3d_impl(OSA_DECONV_ARGS, PREC_F16X3, out_scale, stream, false, nullptr, 0, 0, nullptr, nullptr, nullptr, 1.f, ranges);
}

extern "C" int osa_deconv3d_ndhwc_f16(const void* x_, const float* w_packed, const float* scale, const float* shift, const void* residual_, void* y_,
                                      int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs, int k, int pad, int opad,
                                      const float* gate_logits, int gCs, int act, float slope, void* stream) {
    const float* x = static_cast<const float*>(x_); const float* residual = static_cast<const float*>(residual_); float* y = static_cast<float*>(y_);
    return deconv3d_impl(OSA_DECONV_ARGS, PREC_F16, 1.f, stream);
}

// transposed conv with the 1x1x1 redir branch computed in its epilogue (see ConvArgs::rx)
#define OSA_REDIR_PARAMS const float* rx, int rxCs, int rCi, const float* rw_packed, const float* rscale, const float* rshift
extern "C" int osa_deconv3d_redir_ndhwc_f32(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                            int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs,
                                            int k, int pad, int opad, OSA_REDIR_PARAMS, int act, float slope, void* stream) {
    return deconv3d_impl(x, w_packed, scale, shift, nullptr, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, 0, k, pad, opad, nullptr, 0,
                         act, slope, PREC_F32, 1.f, stream, false, rx, rxCs, rCi, rw_packed, rscale, rshift, 1.f);
}

extern "C" int osa_deconv3d_redir_ndhwc_f16x3(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                              int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs,
                                              int k, int pad, int opad, OSA_REDIR_PARAMS, float r_out_scale,
                                              int act, float slope, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return deconv3d_impl(x, w_packed, scale, shift, nullptr, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, 0, k, pad, opad, nullptr, 0,
                         act, slope, PREC_F16X3, out_scale, stream, false, rx, rxCs, rCi, rw_packed, rscale, rshift, r_out_scale, ranges);
}

#define OSA_DECONV2D_PARAMS                                                                     \
    const float* x, const float* w_packed, const float* scale, const float* shift,             \
    const float* residual, float* y, int B, int Hi, int Wi, int Ci, int xCs,                   \
    int Co, int yCs, int rCs, int k, int pad, int opad, const float* gate_logits, int gCs, int act, float slope
#define OSA_DECONV2D_ARGS                                                                       \
    x, w_packed, scale, shift, residual, y, B, 1, Hi, Wi, Ci, xCs, Co, yCs, rCs, k, pad, opad, \
    gate_logits, gCs, act, slope

extern "C" int osa_deconv2d_nhwc_f32(OSA_DECONV2D_PARAMS, void* stream) {
    return deconv3d_impl(OSA_DECONV2D_ARGS, PREC_F32, 1.f, stream, true);
}

extern "C" int osa_deconv2d_nhwc_f16x3(OSA_DECONV2D_PARAMS, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return deconv3d_impl(OSA_DECONV2D_ARGS, PREC_F16X3, out_scale, stream, true, nullptr, 0, 0, nullptr, nullptr, nullptr, 1.f, ranges);
}

extern "C" int osa_deconv2d_nhwc_f16(const void* x_, const float* w_packed, const float* scale, const float* shift, const void* residual_, void* y_,
                                     int B, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs, int k, int pad, int opad,
                                     const float* gate_logits, int gCs, int act, float slope, void* stream) {
    const float* x = static_cast<const float*>(x_); const float* residual = static_cast<const float*>(residual_); float* y = static_cast<float*>(y_);
    return deconv3d_impl(OSA_DECONV2D_ARGS, PREC_F16, 1.f, stream, true);
}

__global__ __launch_bounds__(256) void small_co_pack_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int Ci, int Co, int T, int nchunks) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // dst index ((ch*T + t)*Co + co)*16 + e
    if (i >= nchunks * T * Co * 16) return;
    const int e = i & 15; int r = i >> 4;
    const int co = r % Co; r /= Co;
    const int t = r % T; const int ch = r / T;
    const int ci = ch * CC + e;
    dst[i] = (ci < Ci) ? src[((size_t)co * Ci + ci) * T + t] : 0.f;
}

extern "C" size_t osa_conv3d_small_co_packed_floats(int Ci, int Co, int kd, int kh, int kw) {
    return (size_t)nchunks_of(Ci) * kd * kh * kw * Co * 16;
}

extern "C" int osa_conv3d_small_co_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                            int kd, int kh, int kw, void* stream) {
    OSA_REQUIRE(w_ref && w_packed, "conv3d_small_co_pack: NULL pointer");
    OSA_REQUIRE(Ci > 0 && Co >= 1 && Co <= 4 && kd > 0 && kh > 0 && kw > 0, "conv3d_small_co_pack: bad dims");
    const int n = (int)osa_conv3d_small_co_packed_floats(Ci, Co, kd, kh, kw);
    hipLaunchKernelGGL(small_co_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_ref, w_packed, Ci, Co, kd * kh * kw, nchunks_of(Ci));
    OSA_LAUNCH_CHECK("conv3d_small_co_pack");
    return 0;
}

static int small_co_impl(const float* x, const float* w, bool packed, const float* bias,
                         const float* residual, float* y,
                         int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                         int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                         void* stream);

extern "C" int osa_conv3d_small_co_ndhwc_f32(const float* x, const float* w_ref, const float* bias,
                                             const float* residual, float* y,
                                             int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                                             int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                                             void* stream) {
    return small_co_impl(x, w_ref, false, bias, residual, y, B, D, H, W, Ci, xCs, Co, yCs, kd, kh, kw, pad_d, pad_h, pad_w, stream);
}

extern "C" int osa_conv3d_small_co_packed_ndhwc_f32(const float* x, const float* w_packed, const float* bias,
                                                    const float* residual, float* y,
                                                    int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                                                    int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                                                    void* stream) {
    OSA_REQUIRE(((size_t)w_packed & 63) == 0, "conv3d_small_co_packed: packed weights must be 64-byte aligned");
    return small_co_impl(x, w_packed, true, bias, residual, y, B, D, H, W, Ci, xCs, Co, yCs, kd, kh, kw, pad_d, pad_h, pad_w, stream);
}

static int small_co_impl(const float* x, const float* w_ref, bool packed, const float* bias,
                         const float* residual, float* y,
                         int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                         int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                         void* stream) {
    OSA_REQUIRE(x && w_ref && y, "conv3d_small_co: NULL pointer");
    OSA_REQUIRE(Co >= 1 && Co <= 4, "conv3d_small_co: Co=%d unsupported (1..4)", Co);
    OSA_REQUIRE(xCs % 4 == 0 && xCs >= Ci && ((size_t)x & 15) == 0, "conv3d_small_co: x must be 16-byte aligned, xCs %% 4 == 0");
    OSA_REQUIRE(kd == 2 * pad_d + 1 && kh == 2 * pad_h + 1 && kw == 2 * pad_w + 1, "conv3d_small_co: only 'same' convolutions");
    OSA_REQUIRE(kd * kh * kw <= MAX_TAPS, "conv3d_small_co: too many taps");
    OSA_REQUIRE(yCs >= Co, "conv3d_small_co: yCs < Co");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.res = residual; a.oscale = 1.f;
    a.B = B; a.Di = D; a.Hi = H; a.Wi = W; a.Ci = Ci; a.xCs = xCs;
    a.Do = D; a.Ho = H; a.Wo = W; a.Co = Co; a.yCs = yCs;
    a.Ad = D; a.Ah = H; a.Aw = W;
    a.isd = a.ish = a.isw = 1; a.os = 1;
    a.T = kd * kh * kw;
    int t = 0;
    for (int z = 0; z < kd; ++z) for (int yy = 0; yy < kh; ++yy) for (int xx = 0; xx < kw; ++xx, ++t) {
        a.td[t] = (signed char)(z - pad_d); a.th[t] = (signed char)(yy - pad_h); a.tw[t] = (signed char)(xx - pad_w);
    }
    a.nchunks = nchunks_of(Ci); a.CoP = Co;
    a.dmin = -pad_d; a.hmin = -pad_h; a.wmin = -pad_w;
    a.LD = 4 + 2 * pad_d; a.LH = 8 + 2 * pad_h; a.LW = 8 + 2 * pad_w;
    a.tilesD = cdiv(D, 4); a.tilesH = cdiv(H, 8); a.tilesW = cdiv(W, 8);
    finish_geometry(a, 8, true);       // thread q reads voxel q of the 4x8x8 tile: the MFMA A-operand pattern, compact image
    hipStream_t st = (hipStream_t)stream;
    if (packed && Co == 1 && Ci == 32 && xCs == 32 && kd == 3 && kh == 3 && kw == 3 && !exp_set("OSA_NO_MARCH")) {
        // the classifier shape: d-marching form (one pass over the input + an 18x18 / 16x16 halo).  D is cut into segments only as far as
        // needed to give every CU several workgroups (each segment re-reads 2 halo planes)
        a.tilesH = cdiv(H, CM_TH); a.tilesW = cdiv(W, CM_TW);
        const long long cols = (long long)B * a.tilesH * a.tilesW;
        int nseg = (int)((2048 + cols - 1) / cols);
        { const int o = exp_int("OSA_MARCH_NSEG", 0); if (o) nseg = o; }
        if (nseg > D / 8) nseg = D / 8;
        if (nseg < 1) nseg = 1;
        const int dseg = cdiv(D, nseg);
        nseg = cdiv(D, dseg);
        OSA_REQUIRE(cols * nseg < (1ll << 31), "conv3d_small_co: grid too large");
        OSA_REQUIRE((long long)H * W * xCs < (1ll << 31), "conv3d_small_co: plane too large");
        const size_t mlds = (size_t)CM_LH * CM_ROWQ * sizeof(float4);
        hipLaunchKernelGGL(classifier_march_kernel, dim3((unsigned)(cols * nseg)), dim3(256), mlds, st, a, w_ref, bias, dseg, nseg);
        OSA_LAUNCH_CHECK("conv3d_small_co (march)");
        return 0;
    }
    const size_t lds = ((size_t)a.LD * a.PlaneQ * 4 + (packed ? 0 : (size_t)a.nchunks * a.T * Co * 16)) * sizeof(float);
    OSA_REQUIRE(lds <= 160 * 1024, "conv3d_small_co: %zu B of LDS needed", lds);
    const long long nblk = (long long)B * a.tilesD * a.tilesH * a.tilesW;
    OSA_REQUIRE(nblk < (1ll << 31), "conv3d_small_co: grid too large");
    dim3 grid((unsigned)nblk), block(256);
#define OSA_SC_LAUNCH1(CO, WG)                                                                              \
    do {                                                                                                    \
        if (lds > 64 * 1024)                                                                                \
            (void)hipFuncSetAttribute((const void*)conv_small_co_tiled_kernel<CO, WG>,                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL((conv_small_co_tiled_kernel<CO, WG>), grid, block, lds, st, a, w_ref, bias);     \
    } while (0)
#define OSA_SC_LAUNCH(CO) do { if (packed) OSA_SC_LAUNCH1(CO, true); else OSA_SC_LAUNCH1(CO, false); } while (0)
    switch (Co) {
        case 1: OSA_SC_LAUNCH(1); break;
        case 2: OSA_SC_LAUNCH(2); break;
        case 3: OSA_SC_LAUNCH(3); break;
        default: OSA_SC_LAUNCH(4); break;
    }
#undef OSA_SC_LAUNCH
    OSA_LAUNCH_CHECK("conv3d_small_co");
    return 0;
}

#ifdef OSA_TRACE_ON
// -DOSA_TRACE_ON builds only: copy the in-kernel timeline (conv_kernel.h, OSA_DBG & 256) to the host and clear it
extern "C" int osa_debug_trace_read(unsigned long long* dst, size_t n_words) {
    const size_t have = (size_t)osa::TRACE_SLOTS * osa::TRACE_WAVES * osa::TRACE_EVENTS;
    if (n_words > have) n_words = have;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(osa::g_trace), n_words * sizeof(unsigned long long)) != hipSuccess) return -1;
    static unsigned long long zeros[osa::TRACE_SLOTS * osa::TRACE_WAVES * osa::TRACE_EVENTS];
    if (hipMemcpyToSymbol(HIP_SYMBOL(osa::g_trace), zeros, sizeof(zeros)) != hipSuccess) return -1;
    return (int)n_words;
}
#endif

extern "C" int osa_conv_b_ring_mask(int mask) { const int prev = osa::g_b_ring_mask; osa::g_b_ring_mask = mask; osa::march_s2_set_waves(((mask >> 28) & 1) ? 4 : 8); osa::wgrad_set_multi_tile((mask >> 27) & 1); return prev; }
extern "C" long long osa_conv_b_ring_launches(void) { return osa::g_b_ring_launches; }
