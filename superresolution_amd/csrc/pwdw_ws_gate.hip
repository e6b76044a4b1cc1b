// Gated instantiation of the wave-specialised fused kernel (see pwdw_ws.hip for why it is a file of its own).
#include "pwdw_ws_kernel.hpp"

namespace rst {

const void* pwdw_ws_gate_kernel(int ks, bool f16s) {
    if (f16s && ks == 24) return reinterpret_cast<const void*>(&pwdw_ws_kernel<24, 8, true, true>);
    return ks == 24 ? reinterpret_cast<const void*>(&pwdw_ws_kernel<24, 8, true>)
                    : reinterpret_cast<const void*>(&pwdw_ws_kernel<12, 8, true>);
}

}  // namespace rst
