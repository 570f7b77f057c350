// TEST INFRASTRUCTURE ONLY. Force-included (-include) when the unmodified
// reference sources under /root/reference/src are compiled into oracle/_ref/.
//
// The reference was written against libc++ (its Makefile passes -stdlib=libc++);
// this image only carries libstdc++ 11, which differs in three ways the
// reference trips over:
//   * `assert` is used (src/vector/vector3d.cpp:8, src/image/PNG.cpp:109) without
//     including <cassert>;
//   * `std::sinf/std::cosf/std::sqrtf` are used (src/bsdf/math_utils.cpp:26-33,
//     src/bsdf/microfacets.cpp:40-44) but libstdc++ 11 declares them only in the
//     global namespace;
//   * libc++'s <cmath> makes the float overloads of sqrt/cos/exp/... visible
//     unqualified; libstdc++ does that only through its <math.h> wrapper.
// Nothing here replaces reference code: it only adds includes / using-declarations.
#pragma once
#ifdef __cplusplus
#include <cassert>
#include <cmath>
#include <math.h>
#include <cstring>
namespace std {
using ::cosf;
using ::sinf;
using ::sqrtf;
}  // namespace std
#endif
