// Host-side fuzz of the workspace arithmetic of libmobocmf_hip (run by tools/asan_host.sh under
// AddressSanitizer + UBSan, CPU only, no GPU call): random valid and invalid layer descriptors -- d 1..33, xdiv 1..49,
// M 1..2048, N' up to 2^20 -- through every *_bytes entry point; for the valid ones every workspace layout is carved out of
// host buffers of EXACTLY the reported sizes (guarded by ASan red zones) and the first / last double of every region is
// written (mobocmf_debug_touch_workspaces); a buffer 256 bytes shorter must be refused.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../include/mobocmf_hip.h"

extern "C" int mobocmf_debug_touch_workspaces(const mobocmf_layer_desc* desc, void* saved, size_t saved_bytes, void* scratch,
                                              size_t scratch_bytes, void* block, size_t block_bytes, void* psaved,
                                              size_t psaved_bytes, void* pscratch, size_t pscratch_bytes, void* cov,
                                              size_t cov_bytes, int32_t* regions);

#define REQUIRE(c)                                                                              \
    do {                                                                                        \
        if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s  (case %d)\n", __FILE__, __LINE__, #c, icase); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const int ncases = argc > 1 ? std::atoi(argv[1]) : 4000;
    const size_t budget = (size_t)3 << 29;      // bytes of host memory one case may touch
    std::mt19937_64 rng(12345);
    auto U = [&](int64_t lo, int64_t hi) { return (int64_t)(lo + rng() % (uint64_t)(hi - lo + 1)); };
    int valid = 0, invalid = 0, touched = 0, skipped_big = 0;
    int64_t regions_total = 0;
    for (int icase = 0; icase < ncases; ++icase) {
        mobocmf_layer_desc d = {};
        d.kind = (int32_t)U(0, 1);
        d.d = (int32_t)U(1, 33);
        d.xdiv = (int32_t)U(1, 49);
        d.M = (int32_t)(U(0, 3) == 0 ? U(1, 2048) : U(1, 300));
        const int64_t nb = U(0, 4) == 0 ? U(1, (1 << 20) / d.xdiv) : U(1, 4096 / d.xdiv + 1);
        d.Np = nb * d.xdiv;
        d.branch = (int32_t)U(0, 1);
        d.want_dx = (int32_t)U(0, 1);
        d.jitter = 1e-6;
        d.min_var = 1e-10;
        d.phase = (int32_t)U(0, 4);
        bool ok = d.d <= MOBOCMF_MAX_D && d.xdiv <= MOBOCMF_MAX_XDIV;
        // the tuning travels in the descriptor: NULL (defaults) or a record with random valid knobs -- the sizes reported and the
        // regions carved below are both derived from it; now and then a record that must be refused
        mobocmf_tuning tn;
        mobocmf_tuning_init(&tn);
        if (U(0, 2) != 0) {
            const int32_t wg[] = {0, 16, 64, 256, 512, 4096}, tr[] = {0, 64, 128}, mw[] = {4, 8, 32};
            tn.syrk_workgroups = wg[U(0, 5)];
            tn.tile_rows = tr[U(0, 2)];
            tn.pair_mode = (int32_t)U(0, 2);
            tn.mid_gemm_waves = mw[U(0, 2)];
            tn.mid_gemm_max = (int32_t)U(0, 4096);
            tn.small_gemm_max = (int32_t)U(1, 512);
            tn.small_panel_max = (int32_t)U(1, 512);
            tn.sparse_backward = (int32_t)U(0, 1);
            tn.potrf_cols = U(0, 1) ? 4 : 1;
            d.tuning = &tn;
            switch (U(0, 15)) {
                case 0: tn.struct_size = 12; ok = false; break;
                case 1: tn.tile_rows = 96; ok = false; break;
                case 2: tn.syrk_workgroups = 8; ok = false; break;
                case 3: tn.potrf_cols = 2; ok = false; break;
                default: break;
            }
        }
        switch (U(0, 11)) {      // corrupt one field now and then
            case 0: d.kind = 2; ok = false; break;
            case 1: d.Np += 1; if (d.Np % d.xdiv) ok = false; break;
            case 2: d.M = 0; ok = false; break;
            case 3: d.phase = 5; ok = false; break;
            case 4: d.branch = 2; ok = false; break;
            case 5: d.xdiv = 0; ok = false; break;
            default: break;
        }
        size_t sv = 0, sc = 0, bb = 0, st = 0, ps = 0, pc = 0, cs = 0, cv = 0;
        const int rc = mobocmf_layer_workspace_bytes(&d, &sv, &sc);
        REQUIRE((rc == MOBOCMF_OK) == ok);
        REQUIRE((mobocmf_chain_block_bytes(&d, &bb, &st) == MOBOCMF_OK) == ok);
        REQUIRE((mobocmf_panel_workspace_bytes(&d, &ps, &pc) == MOBOCMF_OK) == ok);
        REQUIRE((mobocmf_layer_chain_state_bytes(&d, &cs) == MOBOCMF_OK) == ok);
        REQUIRE((mobocmf_predictive_covariance_workspace_bytes(&d, &cv) == MOBOCMF_OK) == ok);
        REQUIRE(mobocmf_layer_workspace_bytes(&d, nullptr, &sc) == MOBOCMF_BAD_ARG);
        if (!ok) { ++invalid; continue; }
        ++valid;
        REQUIRE(cs == st && cs <= sv && st <= bb && ps <= sv);
        {   // syrk workspace: multiples of 128 only
            const int32_t Mr = (int32_t)((d.M + 127) / 128 * 128);
            const int64_t Kd = (d.Np + 127) / 128 * 128;
            size_t sy = 0;
            REQUIRE(mobocmf_syrk_workspace_bytes(Mr, Kd, d.tuning, &sy) == MOBOCMF_OK && sy >= (size_t)Mr * Mr * 8);
            REQUIRE(mobocmf_syrk_workspace_bytes(Mr + 1, Kd, d.tuning, &sy) == MOBOCMF_BAD_ARG);
            REQUIRE(mobocmf_syrk_workspace_bytes(Mr, Kd, d.tuning, nullptr) == MOBOCMF_BAD_ARG);
        }
        if (sv + sc + bb + ps + pc + cv > budget) { ++skipped_big; continue; }
        // uninitialised heap blocks of exactly the reported sizes (ASan red zones on both sides)
        struct Buf { char* p; explicit Buf(size_t n) : p(new char[n ? n : 1]) {} ~Buf() { delete[] p; } char* data() { return p; } };
        Buf saved(sv), scratch(sc), block(bb), psaved(ps), pscratch(pc), cov(cv);
        int32_t regions = 0;
        REQUIRE(mobocmf_debug_touch_workspaces(&d, saved.data(), sv, scratch.data(), sc, block.data(), bb, psaved.data(), ps,
                                               pscratch.data(), pc, cov.data(), cv, &regions) == MOBOCMF_OK);
        regions_total += regions;
        ++touched;
        // a buffer one allocation granule shorter must be refused (the reported sizes are tight)
        REQUIRE(mobocmf_debug_touch_workspaces(&d, saved.data(), sv - 256, scratch.data(), sc, nullptr, 0, nullptr, 0, nullptr, 0,
                                               nullptr, 0, &regions) == MOBOCMF_WORKSPACE_TOO_SMALL);
        REQUIRE(mobocmf_debug_touch_workspaces(&d, nullptr, 0, nullptr, 0, block.data(), bb - 256, nullptr, 0, nullptr, 0,
                                               nullptr, 0, &regions) == MOBOCMF_WORKSPACE_TOO_SMALL);
        REQUIRE(mobocmf_debug_touch_workspaces(&d, nullptr, 0, nullptr, 0, nullptr, 0, psaved.data(), ps - 256, pscratch.data(), pc,
                                               nullptr, 0, &regions) == MOBOCMF_WORKSPACE_TOO_SMALL);
        REQUIRE(mobocmf_debug_touch_workspaces(&d, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, cov.data(), cv - 256,
                                               &regions) == MOBOCMF_WORKSPACE_TOO_SMALL);
    }
    // the one-launch step's descriptor (mobocmf_tiny_model): size queries over random shapes -- the workspace must hold the
    // flat gradient and the documented panel pool, whatever the row counts -- and the step's argument checks, which refuse
    // every descriptor below on the host (no data pointers are set: nothing can be launched from here)
    int tiny_ok = 0, tiny_bad = 0;
    for (int icase = 0; icase < ncases; ++icase) {
        mobocmf_tiny_model t = {};
        t.L = (int32_t)U(-1, 4); t.M = (int32_t)U(-1, 40); t.d = (int32_t)U(-1, 10); t.S = (int32_t)U(-1, 30);
        t.N = (int32_t)U(-5, 1 << 18);
        int32_t r = t.N;
        bool rows_ok = true;
        for (int l = 0; l < MOBOCMF_TINY_MAX_LAYERS; ++l) {
            t.rows[l] = r;
            if (l < t.L && r < 1) rows_ok = false;
            r = (int32_t)U(-2, r > 1 ? r : 1);
        }
        int64_t flat = -1;
        size_t wb = 0;
        const int rc1 = mobocmf_tiny_flat_len(&t, &flat), rc2 = mobocmf_tiny_work_bytes(&t, &wb);
        const bool shape_ok = t.L >= 1 && t.L <= MOBOCMF_TINY_MAX_LAYERS && t.M >= 1 && t.d >= 1;
        REQUIRE((rc1 == MOBOCMF_OK) == shape_ok);
        REQUIRE((rc2 == MOBOCMF_OK) == (shape_ok && t.S >= 1 && rows_ok));
        if (rc2 == MOBOCMF_OK) {
            int64_t pool = 0, cmax = 0;
            for (int l = 0; l < t.L; ++l) {
                const int64_t c = (int64_t)t.rows[l] * (l ? t.S : 1);
                pool += c * (2 * t.M + 11);
                cmax = c > cmax ? c : cmax;
            }
            REQUIRE((int64_t)wb == 8 * (((flat + 1) & ~(int64_t)1) + pool + 3 * (int64_t)(t.M > 8 ? t.M : 8) * cmax + 256 * 17));
            ++tiny_ok;
        } else {
            ++tiny_bad;
        }
        REQUIRE(mobocmf_tiny_elbo_step(&t, &t, 1, 1e-3, 0.9, 0.999, 1e-8, (int32_t)U(-1, 3), nullptr) == MOBOCMF_BAD_ARG);
    }
    std::printf("fuzz_workspaces: mobocmf_tiny_model: %d descriptors sized, %d refused, every step call refused on the host\n",
                tiny_ok, tiny_bad);
    std::printf("fuzz_workspaces: %d cases: %d valid (%d carved out of exact-size host buffers, %lld regions written, %d above the "
                "%zu MB host budget: sizes only), %d invalid (all refused with BAD_ARG) -- no sanitizer report\n",
                ncases, valid, touched, (long long)regions_total, skipped_big, budget >> 20, invalid);
    return 0;
}
