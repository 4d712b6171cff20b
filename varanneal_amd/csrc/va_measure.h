// va_measure.h -- the ONE hook layer for in-kernel measurement.  Product builds define neither macro and every hook
// expands to nothing; the diagnostic libraries (python -m varanneal_amd._build --stamps / --variant pzst -DVA_PZ_STAMPS)
// are never the product.  Ablation builds of earlier rounds (-DVA_E4_ABLATE / VA_E5_ABLATE / VA_NN_ABLATE) were removed
// from the kernels in round 4: their results are kept in profiles/r03_ablation_c3.txt, r03_e5_experiments.txt and
// r03_nnet_c5x_ablation.txt.
//   VA_STAMPS      k_eval4: per-wave wall_clock64 (100 MHz) stamps + HW_REG_HW_ID / HW_REG_XCC_ID into the update-partials
//                  table (tools/timeline.py, tools/timeline2.py)
//   VA_PZ_STAMPS   k_seed: thread 0 of workgroup 0 of seed 0 accumulates the ticks between consecutive marks of the cycle
//                  into pz.stamps (tools/persist_probe.py)
#pragma once

#ifdef VA_STAMPS
#define VA_E4_STAMP_SETUP(dv, w)                                                                                                   \
    unsigned long long *tl = reinterpret_cast<unsigned long long *>((dv).upp) + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10; \
    if ((threadIdx.x & 63) == 0) {                                                                                                 \
        tl[8] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);                                                     \
        tl[9] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned long long)(w);                  \
    }
#define VA_E4_STAMP(i) do { if ((threadIdx.x & 63) == 0) tl[i] = wall_clock64(); } while (0)
#define VA_E4_STAMP_TAIL(last) do { if (last) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl[1] = wall_clock64(); } } while (0)
#else
#define VA_E4_STAMP_SETUP(dv, w)
#define VA_E4_STAMP(i) do { } while (0)
#define VA_E4_STAMP_TAIL(last) do { } while (0)
#endif

#ifdef VA_PZ_STAMPS
#define PZ_MARK_SETUP() long long pz_acc[PZ_NSTAMP - 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pz_prev = wall_clock64(); \
    const long long pz_c0 = clock64(), pz_w0 = pz_prev
#define PZ_MARK(i) do { if (tid == 0) { const long long t_ = wall_clock64(); pz_acc[i] += t_ - pz_prev; pz_prev = t_; } } while (0)
#define PZ_MARK_FLUSH(stamps, cyc) do { if (blockIdx.x == 0 && tid == 0) {                                   \
        pz_acc[9] = clock64() - pz_c0; pz_acc[10] = wall_clock64() - pz_w0;                                  \
        for (int i_ = 0; i_ < PZ_NSTAMP - 1; ++i_) (stamps)[i_] = (double)pz_acc[i_];                        \
        (stamps)[PZ_NSTAMP - 1] = (double)(cyc); } } while (0)
#else
#define PZ_MARK_SETUP() do { } while (0)
#define PZ_MARK(i) do { } while (0)
#define PZ_MARK_FLUSH(stamps, cyc) do { } while (0)
#endif

// VA_FB_STAMPS   k_nnet_fb: thread 0 of one workgroup (block 1 of seed 0) sums the 100 MHz ticks of its phases over the
//                layers -- 0 input image, 1 first product, 2 its barrier, 3 epilogue A, 4 its barrier, 5 second product,
//                6 epilogue B, 7 the whole kernel -- into dv.pz.stamps (read back with va_debug_read_persist)
#ifdef VA_FB_STAMPS
#define FB_MARK_SETUP() long long fb_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fb_prev = wall_clock64(); const long long fb_w0 = fb_prev
#define FB_MARK(i) do { if (tid == 0) { const long long t_ = wall_clock64(); fb_acc[i] += t_ - fb_prev; fb_prev = t_; } } while (0)
#define FB_MARK_FLUSH(stamps) do { if (blockIdx.x == 1 && blockIdx.y == 0 && tid == 0 && (stamps)) {           \
        fb_acc[7] = wall_clock64() - fb_w0;                                                                    \
        for (int i_ = 0; i_ < 8; ++i_) (stamps)[i_] = (double)fb_acc[i_]; } } while (0)
#else
#define FB_MARK_SETUP() do { } while (0)
#define FB_MARK(i) do { } while (0)
#define FB_MARK_FLUSH(stamps) do { } while (0)
#endif
