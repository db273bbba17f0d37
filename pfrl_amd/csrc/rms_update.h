// One element of torch.optim.RMSprop, shared by the optimizer launches (csrc/optim.hip) and the
// optimizer blocks that ride in the last backward launch (csrc/qnet.hip k_conv_wgrad_ride).
#pragma once
#include <hip/hip_runtime.h>

// One element of torch.optim.RMSprop (no momentum), every step one rounded f32 operation in
// the order of torch/optim/rmsprop.py _multi_tensor_rmsprop.
template <bool CENTERED>
__device__ __forceinline__ void rms_update(float &p, float gi, float &sq, float &ga, float lr,
                                           float alpha, float oma, float eps, float weight_decay) {
    if (weight_decay != 0.0f) gi = __fadd_rn(gi, __fmul_rn(weight_decay, p));
    // square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha)
    float s = __fmul_rn(sq, alpha);
    s = __fadd_rn(s, __fmul_rn(oma, __fmul_rn(gi, gi)));
    sq = s;
    float avg;
    if (CENTERED) {
        // grad_avg.lerp_(grad, 1 - alpha)
        float m = ga;
        m = __fadd_rn(m, __fmul_rn(oma, __fsub_rn(gi, m)));
        ga = m;
        // addcmul(square_avg, grad_avg, grad_avg, value=-1).sqrt_().add_(eps)
        avg = __fadd_rn(__fsqrt_rn(__fsub_rn(s, __fmul_rn(m, m))), eps);
    } else {
        avg = __fadd_rn(__fsqrt_rn(s), eps);
    }
    // param.addcdiv_(grad, avg, value=-lr)
    p = __fadd_rn(p, __fmul_rn(-lr, __fdiv_rn(gi, avg)));
}
