// TEMPORARY: entry points declared in spml_hip.h whose kernels are not written yet.
#include "common.cuh"
extern "C" size_t spml_segsort_nll_workspace_bytes(int64_t, int64_t, int) { return 16; }
extern "C" int spml_segsort_nll_fwd_f32(const float*, const int64_t*, const int64_t*, int64_t,
                                        const float*, const int64_t*, int64_t, int, float, int,
                                        float*, float*, void*, size_t, void*) { return SPML_ERR_UNSUPPORTED; }
extern "C" int spml_segsort_nll_bwd_f32(const float*, const int64_t*, const int64_t*, int64_t,
                                        const float*, const int64_t*, int64_t, int, float, int,
                                        const float*, const float*, float*, float*, void*, size_t,
                                        void*) { return SPML_ERR_UNSUPPORTED; }
extern "C" size_t spml_topk_workspace_bytes(int64_t, int64_t, int, int) { return 16; }
extern "C" int spml_topk_affinity_f32(const float*, int64_t, const float*, int64_t, int, int,
                                      const int64_t*, const int64_t*, const uint8_t*, float,
                                      int64_t*, float*, void*, size_t, void*) { return SPML_ERR_UNSUPPORTED; }
