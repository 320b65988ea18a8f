// fovraster -- the three parameter activations of a 3DGS model in one pass each way.
//
// Reference behaviour (fov3dgs/scene/gaussian_model.py:200-240, torch ops over all P Gaussians, every iteration):
//   get_scaling  = exp(_scaling)            [P,3]
//   get_rotation = normalize(_rotation)     [P,4]   x / max(|x|_2, 1e-12)
//   get_opacity  = sigmoid(_opacity)        [P,1]
// In torch that is ~8 kernels forward (two of them row reductions over [P,4]) and ~10 backward: 0.45 ms per training
// iteration at 6 M Gaussians. Here: one streaming kernel forward (32 B read, 32 B written per Gaussian) and one backward.
#include "common.h"

namespace fr {

__global__ void __launch_bounds__(256) k_activate_fwd(int P, const float *__restrict__ rs, const float *__restrict__ rq, const float *__restrict__ ro,
	float *__restrict__ s, float *__restrict__ q, float *__restrict__ o)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= (size_t)P) return;
#pragma unroll
	for (int k = 0; k < 3; k++) s[3 * i + k] = act_scale(rs[3 * i + k]);
	((float4 *)q)[i] = act_rotation(((const float4 *)rq)[i]);
	o[i] = act_opacity(ro[i]);
}

// gs / gq / go: gradients w.r.t. the activated values (any may be null = zero)
__global__ void __launch_bounds__(256) k_activate_bwd(int P, const float *__restrict__ rs, const float *__restrict__ rq, const float *__restrict__ ro,
	const float *__restrict__ gs, const float *__restrict__ gq, const float *__restrict__ go,
	float *__restrict__ ds, float *__restrict__ dq, float *__restrict__ dop)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= (size_t)P) return;
#pragma unroll
	for (int k = 0; k < 3; k++) ds[3 * i + k] = gs ? gs[3 * i + k] * expf(rs[3 * i + k]) : 0.0f;
	float4 r = make_float4(0, 0, 0, 0);
	if (gq)
	{
		const float4 v = ((const float4 *)rq)[i], g = ((const float4 *)gq)[i];
		const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
		if (n > 1e-12f)
		{
			const float inv = 1.0f / n;
			const float4 u = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
			const float dot = u.x * g.x + u.y * g.y + u.z * g.z + u.w * g.w;
			r = make_float4((g.x - u.x * dot) * inv, (g.y - u.y * dot) * inv, (g.z - u.z * dot) * inv, (g.w - u.w * dot) * inv);
		}
		else r = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f); // clamped denominator: x / 1e-12
	}
	((float4 *)dq)[i] = r;
	const float sg = 1.0f / (1.0f + expf(-ro[i]));
	dop[i] = go ? go[i] * sg * (1.0f - sg) : 0.0f;
}

int launch_activate_forward(int P, const float *rs, const float *rq, const float *ro, float *s, float *q, float *o, hipStream_t stream)
{
	hipLaunchKernelGGL(k_activate_fwd, dim3((unsigned)(((size_t)P + 255) / 256)), dim3(256), 0, stream, P, rs, rq, ro, s, q, o);
	return check_launch("activate_forward", stream, false);
}

int launch_activate_backward(int P, const float *rs, const float *rq, const float *ro, const float *gs, const float *gq, const float *go,
	float *ds, float *dq, float *dop, hipStream_t stream)
{
	hipLaunchKernelGGL(k_activate_bwd, dim3((unsigned)(((size_t)P + 255) / 256)), dim3(256), 0, stream, P, rs, rq, ro, gs, gq, go, ds, dq, dop);
	return check_launch("activate_backward", stream, false);
}

} // namespace fr
