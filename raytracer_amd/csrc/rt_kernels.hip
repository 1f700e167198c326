// rt_kernels.hip — one GROUP of render_kernel instantiations per translation unit.
//
// The library holds 80 instantiations of render_kernel (rt_kernel.h); compiled in one file they took five minutes
// on one core.  This file is compiled once per group, -DRT_KGROUP=g with
//     g = 4 * (f64 chain) + 2 * (lane-mode walks) + (path code: some material spawns rays)
// and exports kernel_group_<g>(variant): the kernel of `variant` = 8 * counters + 4 * cylinder/triangle code +
// 2 * hierarchy/gate code + area-light code, or nullptr where the group has none.  Lane mode only exists for the
// hierarchy variants and always carries the path code, so the groups are 0, 1, 3, 4, 5, 7 (rt_amd.hip picks).
#include "rt_kernel.h"

#ifndef RT_KGROUP
#error "compile with -DRT_KGROUP=0|1|3|4|5|7"
#endif

namespace rtk {

using KernelFn = void (*)(const KParams);

namespace {
constexpr bool kX = (RT_KGROUP & 4) != 0, kL = (RT_KGROUP & 2) != 0, kP = (RT_KGROUP & 1) != 0;
static_assert(!kL || kP, "lane-mode kernels always hold the path code");

// (timing experiments on ONE family, scripts/exp_build.sh: -DRT_EXP_ONLY_TREE compiles the plain hierarchy kernels only)
template <bool S, bool E, bool T, bool A>
constexpr KernelFn pick() {
#ifdef RT_EXP_ONLY_TREE
    return render_kernel<S, false, true, false, kX, kL, true>;
#elif defined(RT_EXP_ONLY_FLAT_AREA)  // (the headline's kernels only)
    return render_kernel<S, false, false, true, kX, false, kP>;
#elif defined(RT_EXP_ONLY_TREE_AREA)  // (cover_image_area's kernels only)
    return render_kernel<S, false, true, true, kX, false, true>;
#elif defined(RT_EXP_ONLY_TREE_EXOTIC)  // (obj_mesh's and the triangle soups' kernels only)
    return render_kernel<S, true, true, false, kX, kL, true>;
#else
    if constexpr (kL && !T) return nullptr;  // lane mode needs a hierarchy
    else return render_kernel<S, E, T, A, kX, kL, kP>;
#endif
}
}  // namespace

#define RT_CAT2(a, b) a##b
#define RT_CAT(a, b) RT_CAT2(a, b)
KernelFn RT_CAT(kernel_group_, RT_KGROUP)(int variant) {
    static const KernelFn table[16] = {
        pick<false, false, false, false>(), pick<false, false, false, true>(), pick<false, false, true, false>(), pick<false, false, true, true>(),
        pick<false, true, false, false>(),  pick<false, true, false, true>(),  pick<false, true, true, false>(),  pick<false, true, true, true>(),
        pick<true, false, false, false>(),  pick<true, false, false, true>(),  pick<true, false, true, false>(),  pick<true, false, true, true>(),
        pick<true, true, false, false>(),   pick<true, true, false, true>(),   pick<true, true, true, false>(),   pick<true, true, true, true>()};
    return (variant >= 0 && variant < 16) ? table[variant] : nullptr;
}

}  // namespace rtk
