// d-marching 3x3x3 convolutions (conv_march.h): instantiations and host-side launch.  A translation unit of its own so that the
// brick kernel's ~200 instantiations (conv3d.hip) are not recompiled with it.
#include "conv_march.h"
#include "conv_march_s2.h"

namespace osa {

// ------------------------------------------------------------------ d-marching form (conv_march.h) --
// 3x3x3 stride-1 "same" convolutions with 32 output channels in the f16x3 mode.  Returns 1 when launched, 0 when the layer is not
// eligible (the brick kernel then runs it), -1 on error.
static long long g_march_launches = 0;
long long march_launches() { return g_march_launches; }

struct MarchCfg { int nwv, tw, th; size_t lds; void (*fn[2][2])(const ConvArgs, int, int); };   // [split output][split input]
#define OSA_MARCH_CFG(NWV, TW) { NWV, TW, MarchGeo<NWV, TW>::TH, MarchGeo<NWV, TW>::lds_bytes(),                      \
                                 { { conv_march_kernel<NWV, TW, 0, 0>, conv_march_kernel<NWV, TW, 0, 1> },            \
                                   { conv_march_kernel<NWV, TW, 1, 0>, conv_march_kernel<NWV, TW, 1, 1> } } }
static const MarchCfg g_march_cfgs[] = {
    OSA_MARCH_CFG(4, 16),     // 0: 16 x 16 pixel column, 4 waves, 2 workgroups per CU
    OSA_MARCH_CFG(4, 32),     // 1:  8 x 32
};

int launch_conv_march(ConvArgs& a, hipStream_t st, const char* what) {
    if (!exp_int("OSA_MARCH", 1)) return 0;
    // (argument validation shared with the brick form -- check_common, check_split_ranges -- has run in conv3d_impl; what follows is
    // ELIGIBILITY: a layer this form does not cover falls through to the brick kernel, it is not an error)
    if (a.T != 27 || a.Co != 32 || a.CoP != 32 || a.Di < 3 || a.gate || a.rx) return 0;
    if (a.Ci % 16 != 0) return 0;                          // whole 16-channel chunks (the staging reads 16-channel rows, the split form is per chunk)
    const int actk = a.act & 15;
    if (actk > OSA_ACT_LEAKY || (a.act & (OSA_GATE_RAW | OSA_RES_AFTER_ACT)) || ((unsigned)a.act >> 16)) return 0;
    if ((a.yCs & 3) || ((size_t)a.y & 15) || (a.res && ((a.rCs & 3) || ((size_t)a.res & 15)))) return 0;
    if ((long long)a.Do * a.Ho * a.Wo * (a.yCs > a.rCs ? a.yCs : a.rCs) >= (1ll << 31)) return 0;
    if ((long long)a.Di * a.Hi * a.Wi * a.xCs >= (1ll << 31)) return 0;
    if (a.act & OSA_IN_SPLIT) OSA_REQUIRE(a.Ci % 16 == 0, "%s: split input needs Ci %% 16 == 0 (got %d)", what, a.Ci);
    if (a.act & OSA_OUT_SPLIT) {
        OSA_REQUIRE(a.yCs % 16 == 0, "%s: split output needs yCs %% 16 == 0", what);
        if (a.res) OSA_REQUIRE(a.act & OSA_RES_SPLIT, "%s: a split output takes a split residual", what);
    }
    if ((a.act & OSA_RES_SPLIT) && a.res) OSA_REQUIRE(a.rCs % 16 == 0, "%s: split residual needs rCs %% 16 == 0", what);
    // pixel-column shape: least padded area among the 4-wave columns (16 x 16 first: measured 2-4 % ahead of 8 x 32 at 544x960,
    // profiles/round4/march_v3_ring4_ablation_ab.txt; the 2-wave 8 x 16 column loses 20 % since the B ring costs it a workgroup per CU)
    int gi = exp_int("OSA_MARCH_GEO", -1);
    if (gi < 0 || gi > 1) {
        long long best = -1;
        for (int i = 0; i < 2; ++i) {
            const MarchCfg& g = g_march_cfgs[i];
            const long long area = (long long)cdiv(a.Ho, g.th) * g.th * cdiv(a.Wo, g.tw) * g.tw;
            if (best < 0 || area < best) { best = area; gi = i; }
        }
    }
    const MarchCfg& g = g_march_cfgs[gi];
    a.tilesD = 1; a.tilesH = cdiv(a.Ho, g.th); a.tilesW = cdiv(a.Wo, g.tw);
    a.dmin = a.hmin = a.wmin = -1;
    a.LD = 1; a.LH = g.th + 2; a.LW = g.tw + 2;
    a.VQ = 5;
    a.RowQ = (g.tw == 16) ? ((a.LW * 5 + 15) / 16 * 16) : a.LW * 5;
    a.PlaneQ = a.LH * a.RowQ;          // (one plane per pass: LD = 1; only the fp32-input variant stages through ConvArgs geometry)
    a.magicW = (unsigned)((0x100000000ull + a.LW - 1) / a.LW);
    a.magicH = (unsigned)((0x100000000ull + a.LH - 1) / a.LH);
    a.magicHW = (unsigned)((0x100000000ull + (unsigned long long)a.LH * a.LW - 1) / ((unsigned long long)a.LH * a.LW));
    a.cps = 1; a.dma = 0; a.dbg = exp_int("OSA_DBG", 0);
    // D segments: a workgroup walks dseg output planes (+ 2 boundary planes that are staged for one third of their taps).  Cost model in
    // plane-steps: rounds of resident workgroups x (dseg + 2 boundary planes of a cut column, staged and multiplied in full); the fewest segments win a tie.
    const long long cols = (long long)a.B * a.tilesH * a.tilesW;
    int per_cu = (int)((160 * 1024) / g.lds);                       // resident workgroups per CU: LDS, and 2 waves per SIMD by registers
    if (per_cu > 8 / g.nwv) per_cu = 8 / g.nwv;
    const long long slots = (long long)per_cu * 256;
    int nseg = 1;
    {
        double best = 1e30;
        for (int n = 1; n <= 16 && n * 2 <= a.Di; ++n) {
            const int ds = cdiv(a.Di, n), ns = cdiv(a.Di, ds);
            if (ns != n) continue;
            const double cost = (double)((cols * ns + slots - 1) / slots) * (ds + (ns > 1 ? 2.0 : 0.0));
            if (cost < best - 1e-9) { best = cost; nseg = ns; }
        }
        const int o = exp_int("OSA_MARCH_NSEG", 0);
        if (o > 0 && o * 2 <= a.Di) nseg = o;
    }
    const int dseg = cdiv(a.Di, nseg);
    nseg = cdiv(a.Di, dseg);
    OSA_REQUIRE(cols * nseg < (1ll << 31), "%s: grid too large", what);
    void (*fn)(const ConvArgs, int, int) = g.fn[(a.act & OSA_OUT_SPLIT) ? 1 : 0][(a.act & OSA_IN_SPLIT) ? 1 : 0];
    static bool attr_set[2][2][2];
    bool& done = attr_set[gi][(a.act & OSA_OUT_SPLIT) ? 1 : 0][(a.act & OSA_IN_SPLIT) ? 1 : 0];
    if (!done && g.lds > 64 * 1024) { (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); }
    done = true;
    hipLaunchKernelGGL(fn, dim3((unsigned)(cols * nseg)), dim3(g.nwv * 64), g.lds, st, a, dseg, nseg);
    OSA_LAUNCH_CHECK(what);
    ++g_march_launches;
    return 1;
}

// ------------------------------------------------------------------ stride-2 d-marching form (conv_march_s2.h) --
// 3x3x3 stride-2 pad-1 convolutions with 64 output channels, f16x3 mode, split input AND split output, no residual (conv1 of the GwcNet /
// PSMNet hourglasses).  Returns 1 when launched, 0 when the layer is not eligible (the brick kernel then runs it), -1 on error.
static long long g_march_s2_launches = 0;
long long march_s2_launches() { return g_march_s2_launches; }

static int g_march_s2_waves = 8;                            // 8: 4 x 32 columns, one workgroup per CU (measured ahead: profiles/round6/march_s2_versions.txt); 4: 2 x 32 columns, two per CU (osa_conv_b_ring_mask bit 28)
template <int NWV>
static int launch_conv_march_s2_t(ConvArgs& a, hipStream_t st, const char* what) {
    using G = MarchS2Geo<NWV>;
    a.tilesD = 1; a.tilesH = cdiv(a.Ho, G::TH); a.tilesW = cdiv(a.Wo, G::TW);
    a.dbg = exp_int("OSA_DBG", 0);
    // D segments of `oseg` output planes: a segment stages 2 oseg + 1 input planes (the first one for a third of its taps).  Cost model in
    // plane-steps per round of resident workgroups; the fewest segments win a tie.
    const long long cols = (long long)a.B * a.tilesH * a.tilesW;
    const long long slots = 256 * (NWV == 8 ? 1 : 2);
    int nseg = 1;
    {
        double best = 1e30;
        for (int n = 1; n <= a.Do; ++n) {
            const int os = cdiv(a.Do, n), ns = cdiv(a.Do, os);
            if (ns != n) continue;
            const double cost = (double)((cols * ns + slots - 1) / slots) * (2.0 * os + (ns > 1 ? 1.0 : 0.0) + 0.5);     // (+ 0.5: prologue / drain of a workgroup)
            if (cost < best - 1e-9) { best = cost; nseg = ns; }
        }
        const int o = exp_int("OSA_MARCH_NSEG", 0);
        if (o > 0 && o <= a.Do) nseg = o;
    }
    const int oseg = cdiv(a.Do, nseg);
    nseg = cdiv(a.Do, oseg);
    OSA_REQUIRE(cols * nseg < (1ll << 31), "%s: grid too large", what);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_march_s2_kernel<NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes()); attr_set = true; }
    hipLaunchKernelGGL(conv_march_s2_kernel<NWV>, dim3((unsigned)(cols * nseg)), dim3(NWV * 64), G::lds_bytes(), st, a, oseg, nseg);
    OSA_LAUNCH_CHECK(what);
    return 1;
}

void march_s2_set_waves(int w) { g_march_s2_waves = (w == 4) ? 4 : 8; }

int launch_conv_march_s2(ConvArgs& a, hipStream_t st, const char* what) {
    if (a.T != 27 || a.Co != 64 || a.CoP != 64 || a.Di < 2 || a.gate || a.rx || a.res) return 0;
    if (a.Ci % 16 != 0 || a.xCs % 16 != 0) return 0;
    if (!(a.act & OSA_IN_SPLIT) || !(a.act & OSA_OUT_SPLIT)) return 0;
    const int actk = a.act & 15;
    if (actk > OSA_ACT_LEAKY || (a.act & (OSA_GATE_RAW | OSA_RES_AFTER_ACT)) || ((unsigned)a.act >> 16)) return 0;
    if ((a.yCs % 16) || ((size_t)a.y & 15) || ((size_t)a.x & 15)) return 0;
    if ((long long)a.Do * a.Ho * a.Wo * a.yCs >= (1ll << 31)) return 0;
    if ((long long)a.Hi * a.Wi * a.xCs >= (1ll << 29)) return 0;        // per-plane byte offsets are 32-bit
    const int r = g_march_s2_waves == 8 ? launch_conv_march_s2_t<8>(a, st, what) : launch_conv_march_s2_t<4>(a, st, what);
    if (r == 1) ++g_march_s2_launches;
    return r;
}

}  // namespace osa

extern "C" long long osa_conv3d_march_launches(void) { return osa::march_launches(); }
extern "C" long long osa_conv3d_march_s2_launches(void) { return osa::march_s2_launches(); }
