// Entry points not implemented yet return LNASR_E_UNSUPPORTED (they never fall back to the CPU).
#include "common.h"
using namespace lnasr;
#define LN_TODO(name) return fail(LNASR_E_UNSUPPORTED, name " is not implemented yet")
extern "C" {
int lnasr_estimate_discrete(lnasr_ctx*, const int32_t*, int, int, int, int, const void*, const void*, void*, void*, void*) { LN_TODO("lnasr_estimate_discrete"); }
int lnasr_estimate_gmm(lnasr_ctx*, const double*, int, int, const void*, const void*, const void*, const void*, const void*, double, void*, void*, void*, double*, double*) { LN_TODO("lnasr_estimate_gmm"); }
int lnasr_bw_stats_len(lnasr_ctx*, size_t*) { LN_TODO("lnasr_bw_stats_len"); }
int lnasr_bw_accumulate(lnasr_ctx*, const double*, int, int, const int32_t*, double*) { LN_TODO("lnasr_bw_accumulate"); }
int lnasr_bw_mstep(lnasr_ctx*, const double*, double, double*, double*, double*, double*, double*) { LN_TODO("lnasr_bw_mstep"); }
int lnasr_lse(lnasr_ctx*, const double*, size_t, size_t, size_t, int, double*) { LN_TODO("lnasr_lse"); }
int lnasr_num_frames(int64_t, int, int, int64_t*) { LN_TODO("lnasr_num_frames"); }
int lnasr_split_frames(lnasr_ctx*, const double*, int64_t, int, int, double*) { LN_TODO("lnasr_split_frames"); }
int lnasr_mfcc(lnasr_ctx*, const void*, int, const int64_t*, int, const lnasr_mfcc_cfg*, const double*, const double*, double*, double*, double*) { LN_TODO("lnasr_mfcc"); }
}
