// zpn_variant.hip -- the headline's instance of the one-block zero-phase chain kernel
// (chain_zpn_kernel<27, 6, 2>: 1024 taps, the 6-section band-pass) ALONE, as the translation unit
// that replaces csrc/chain_zpn_6.hip in a variant build of the library: seconds to compile instead
// of two minutes, so that kernel variants (-D flags) can be timed against each other on one box
// with benchmarks/ab_chain.cpp.  benchmarks/build_variant.sh NAME [-D...] links
// benchmarks/bin/lib_NAME.so from it and the library's other objects.
//   -DOSZ_ABL_ANY -DOSZ_ABL_NOEPI / NOSTORE / NOH / NODMA / NOLDS   what a part of the kernel costs
//   -DOSZ_CLK                                                        shader clock of a workgroup's life
//   -DOSZ_NO_NT                                                      rows without the non-temporal hint
#define OSZ_ZPN_NM 6
#define OSZ_ZPN_NO_DISPATCH
#include "../openseize_amd/csrc/chain_zpn_body.h"
namespace osz {
zp_kern_t zpn_kernel_nm6(int nb, int ns, int r) { return (nb == 27 && ns == 2 && r <= 5) ? chain_zpn_kernel<27, 6, 2> : nullptr; }
zp_kern_t zpn_fwd_kernel_nm6(int, int) { return nullptr; }
}  // namespace osz
#ifdef OSZ_CLK
extern "C" int osz_dbg_clk(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(osz::g_zpn_clk), 32); }
#endif
