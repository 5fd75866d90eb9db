// Every instantiation of the heavy kernel templates of kernels.h, grouped by the translation unit that compiles it
// (csrc/hip/inst_*.hip define SPT_INSTANTIATE_GROUP_x and get explicit instantiation DEFINITIONS; spt_hip.hip, which
// only launches them, gets explicit instantiation DECLARATIONS so that it compiles none of them).  One monolithic
// .hip took ~4 minutes per library build; split this way the groups compile side by side.
#pragma once
#include "kernels.h"

#define SPT_UNPAREN(...) __VA_ARGS__
#define SPT_DECLARE_KERNEL(NAME, ARGS) extern template __global__ void SPT_UNPAREN NAME ARGS;
#define SPT_DEFINE_KERNEL(NAME, ARGS) template __global__ void SPT_UNPAREN NAME ARGS;

#define SPT_ARGS_PRIMARY (DScene, RenderCtx)
#define SPT_ARGS_BOUNCE (DScene, RenderCtx, uint32_t)

// k_primary<kLds, kChunked, kCount, kEye>
#define SPT_KERNELS_PRIMARY(X)                          \
    X((k_primary<true, false, false>), SPT_ARGS_PRIMARY)  \
    X((k_primary<true, true, false>), SPT_ARGS_PRIMARY)   \
    X((k_primary<true, false, false, true>), SPT_ARGS_PRIMARY)  \
    X((k_primary<true, true, false, true>), SPT_ARGS_PRIMARY)   \
    X((k_primary<false, false, false>), SPT_ARGS_PRIMARY) \
    X((k_primary<false, true, false>), SPT_ARGS_PRIMARY)  \
    X((k_primary<false, false, true>), SPT_ARGS_PRIMARY)  \
    X((k_primary<false, true, true>), SPT_ARGS_PRIMARY)

// k_shadow / k_extend <kLds, kCount, kFlat>, refilling variants <kCount>
#define SPT_KERNELS_RAYS(X)                          \
    X((k_shadow<true, false>), SPT_ARGS_BOUNCE)      \
    X((k_shadow<true, false, true>), SPT_ARGS_BOUNCE) \
    X((k_shadow<false, false>), SPT_ARGS_BOUNCE)     \
    X((k_shadow<false, true>), SPT_ARGS_BOUNCE)      \
    X((k_extend<true, false>), SPT_ARGS_BOUNCE)      \
    X((k_extend<true, false, true>), SPT_ARGS_BOUNCE) \
    X((k_extend<false, false>), SPT_ARGS_BOUNCE)     \
    X((k_extend<false, true>), SPT_ARGS_BOUNCE)      \
    X((k_shadow_dyn<false>), SPT_ARGS_BOUNCE)        \
    X((k_shadow_dyn<true>), SPT_ARGS_BOUNCE)         \
    X((k_extend_dyn<false>), SPT_ARGS_BOUNCE)        \
    X((k_extend_dyn<true>), SPT_ARGS_BOUNCE)

// streaming kernels (stream.h): k_shadow_stream / k_extend_stream <kCount>, k_primary_stream<kChunked, kCount>
#define SPT_KERNELS_STREAM(X)                             \
    X((k_shadow_stream<false>), SPT_ARGS_BOUNCE)          \
    X((k_shadow_stream<true>), SPT_ARGS_BOUNCE)           \
    X((k_extend_stream<false>), SPT_ARGS_BOUNCE)          \
    X((k_extend_stream<true>), SPT_ARGS_BOUNCE)           \
    X((k_primary_stream<false, false>), SPT_ARGS_PRIMARY) \
    X((k_primary_stream<true, false>), SPT_ARGS_PRIMARY)  \
    X((k_primary_stream<false, true>), SPT_ARGS_PRIMARY)  \
    X((k_primary_stream<true, true>), SPT_ARGS_PRIMARY)

// k_shade<kFeat, kFirst, kFused, kTab, kGeoLds>
#define SPT_KERNELS_SHADE0(X)                                        \
    X((k_shade<0, true, true, true, true>), SPT_ARGS_BOUNCE)         \
    X((k_shade<0, false, true, true, true>), SPT_ARGS_BOUNCE)        \
    X((k_shade<0, false, true, true, true, true>), SPT_ARGS_BOUNCE)  \
    X((k_shade<0, true, false, true, true>), SPT_ARGS_BOUNCE)        \
    X((k_shade<0, false, false, true, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<0, true, false, false, false>), SPT_ARGS_BOUNCE)      \
    X((k_shade<0, false, false, false, false>), SPT_ARGS_BOUNCE)
#define SPT_KERNELS_SHADE1(X)                                        \
    X((k_shade<1, true, false, true, true>), SPT_ARGS_BOUNCE)        \
    X((k_shade<1, false, false, true, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<1, true, false, false, false>), SPT_ARGS_BOUNCE)      \
    X((k_shade<1, false, false, false, false>), SPT_ARGS_BOUNCE)
#define SPT_KERNELS_SHADE2(X)                                        \
    X((k_shade<2, true, false, true, true>), SPT_ARGS_BOUNCE)        \
    X((k_shade<2, false, false, true, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<2, true, false, false, false>), SPT_ARGS_BOUNCE)      \
    X((k_shade<2, false, false, false, false>), SPT_ARGS_BOUNCE)
#define SPT_KERNELS_SHADE3A(X)                                       \
    X((k_shade<3, true, false, true, true>), SPT_ARGS_BOUNCE)        \
    X((k_shade<3, false, false, true, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<3, true, false, false, false>), SPT_ARGS_BOUNCE)
#define SPT_KERNELS_SHADE3B(X)                                       \
    X((k_shade<3, false, false, false, false>), SPT_ARGS_BOUNCE)     \
    X((k_shade<3, true, false, false, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<3, false, false, false, true>), SPT_ARGS_BOUNCE)

// k_shade<4, .>: position-normal distributions without a BSSRDF probe (kGeoLds plays no role: two table variants)
#define SPT_KERNELS_SHADE4(X)                                        \
    X((k_shade<4, true, false, true, true>), SPT_ARGS_BOUNCE)        \
    X((k_shade<4, false, false, true, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<4, true, false, false, false>), SPT_ARGS_BOUNCE)      \
    X((k_shade<4, false, false, false, false>), SPT_ARGS_BOUNCE)
// k_shade<5, .>: both
#define SPT_KERNELS_SHADE5A(X)                                       \
    X((k_shade<5, true, false, true, true>), SPT_ARGS_BOUNCE)        \
    X((k_shade<5, false, false, true, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<5, true, false, false, false>), SPT_ARGS_BOUNCE)
#define SPT_KERNELS_SHADE5B(X)                                       \
    X((k_shade<5, false, false, false, false>), SPT_ARGS_BOUNCE)     \
    X((k_shade<5, true, false, false, true>), SPT_ARGS_BOUNCE)       \
    X((k_shade<5, false, false, false, true>), SPT_ARGS_BOUNCE)

#if defined(SPT_INSTANTIATE_GROUP_PRIMARY)
SPT_KERNELS_PRIMARY(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_RAYS)
SPT_KERNELS_RAYS(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_STREAM)
SPT_KERNELS_STREAM(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE0)
SPT_KERNELS_SHADE0(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE1)
SPT_KERNELS_SHADE1(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE2)
SPT_KERNELS_SHADE2(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE3A)
SPT_KERNELS_SHADE3A(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE3B)
SPT_KERNELS_SHADE3B(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE4)
SPT_KERNELS_SHADE4(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE5A)
SPT_KERNELS_SHADE5A(SPT_DEFINE_KERNEL)
#elif defined(SPT_INSTANTIATE_GROUP_SHADE5B)
SPT_KERNELS_SHADE5B(SPT_DEFINE_KERNEL)
#else
SPT_KERNELS_PRIMARY(SPT_DECLARE_KERNEL)
SPT_KERNELS_RAYS(SPT_DECLARE_KERNEL)
SPT_KERNELS_STREAM(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE0(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE1(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE2(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE3A(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE3B(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE4(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE5A(SPT_DECLARE_KERNEL)
SPT_KERNELS_SHADE5B(SPT_DECLARE_KERNEL)
#endif
