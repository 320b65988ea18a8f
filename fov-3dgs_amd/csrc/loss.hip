// fovraster -- fused L1 + SSIM image loss (the step that follows the rasterizer in a training
// iteration, eff_finetune.py:124-125).
//
// Reference behaviour (fov3dgs/utils/loss_utils.py):
//   l1_loss :17-18   mean |x - y|
//   ssim    :37-76   per channel: mu = conv2d(img, w), E = conv2d(img*img', w) with the 11x11 Gaussian window
//                    (sigma 1.5, outer product of the normalised 1-D window, :26-34), zero padding 5;
//                    sigma = E - mu*mu; map = (2 mu1 mu2 + C1)(2 sigma12 + C2) / ((mu1^2 + mu2^2 + C1)(sigma1 + sigma2 + C2)),
//                    C1 = 0.01^2, C2 = 0.03^2; mean over all elements.
// The reference runs five grouped conv2d's of 121 taps each plus ~15 elementwise kernels, and autograd runs the
// transposed convolutions again. Here one kernel per direction does everything on a 16x32 tile staged in LDS with its
// 5-pixel halo: the window is separable, so a tile costs 2 x 11 taps per quantity instead of 121, and the images are
// read once. The forward leaves three derivative maps per channel (with mu, sigma as functions of the image x:
// d map / d x(q) = w(p - q) [A(p) + 2 x(q) B(p) + y(q) C(p)]), so the backward is one more separable pass:
//   dL/dx = w_l1 sign(x - y) + w_ssim (conv(A) + 2 x conv(B) + y conv(C)).
// Sums are written per workgroup (no float atomics: the loss is reproducible run to run).
#include "common.h"

namespace fr {

#define FR_LOSS_TW 16                              // tile: 16 pixels wide, 32 tall: every thread owns two vertically
#define FR_LOSS_TH 32                              // adjacent outputs, so the 11-tap column pass reads 12 values for two
#define FR_LOSS_HALO 5                             // outputs and the row pass 12 + 12 for two adjacent columns -- half the
#define FR_LOSS_SW (FR_LOSS_TW + 2 * FR_LOSS_HALO) // LDS reads of one output per thread (the kernels are bound by LDS reads)
#define FR_LOSS_SH (FR_LOSS_TH + 2 * FR_LOSS_HALO)

// normalised 1-D window gaussian(11, 1.5) of loss_utils.py:26-28
__device__ __constant__ float c_win[11] = {
	1.0283801239e-03f, 7.5987582095e-03f, 3.6000773311e-02f, 1.0936068743e-01f, 2.1300552785e-01f, 2.6601171494e-01f,
	2.1300552785e-01f, 1.0936068743e-01f, 3.6000773311e-02f, 7.5987582095e-03f, 1.0283801239e-03f };

__global__ void __launch_bounds__(256) k_l1_ssim_fwd(int C, int H, int W, const float *__restrict__ x, const float *__restrict__ y,
	float *__restrict__ dmaps, float *__restrict__ partials)
{
	__shared__ float sx[FR_LOSS_SH][FR_LOSS_SW + 1], sy[FR_LOSS_SH][FR_LOSS_SW + 1];
	__shared__ float hs[5][FR_LOSS_SH][FR_LOSS_TW + 1];
	__shared__ float s_red[2][4];
	const int tid = threadIdx.x, tx = tid & 15, ty2 = tid >> 4;
	const int bx = blockIdx.x * FR_LOSS_TW, by = blockIdx.y * FR_LOSS_TH, ch = blockIdx.z;
	const size_t plane = (size_t)H * W;
	const float *X = x + ch * plane, *Y = y + ch * plane;
	for (int i = tid; i < FR_LOSS_SH * FR_LOSS_SW; i += 256)
	{
		const int r = i / FR_LOSS_SW, c = i - r * FR_LOSS_SW;
		const int gy = by + r - FR_LOSS_HALO, gx = bx + c - FR_LOSS_HALO;
		const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
		sx[r][c] = in ? X[(size_t)gy * W + gx] : 0.0f;
		sy[r][c] = in ? Y[(size_t)gy * W + gx] : 0.0f;
	}
	__syncthreads();
	// rows: 42 x 8 pairs of adjacent columns, five running sums per column
	for (int i = tid; i < FR_LOSS_SH * (FR_LOSS_TW / 2); i += 256)
	{
		const int r = i >> 3, c = (i & 7) * 2;
		float a[12], b[12];
#pragma unroll
		for (int k = 0; k < 12; k++) { a[k] = sx[r][c + k]; b[k] = sy[r][c + k]; }
#pragma unroll
		for (int o = 0; o < 2; o++)
		{
			float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
			for (int k = 0; k < 11; k++)
			{
				const float w = c_win[k], av = a[k + o], bv = b[k + o];
				m1 = fmaf(w, av, m1); m2 = fmaf(w, bv, m2);
				e11 = fmaf(w, av * av, e11); e22 = fmaf(w, bv * bv, e22); e12 = fmaf(w, av * bv, e12);
			}
			hs[0][r][c + o] = m1; hs[1][r][c + o] = m2; hs[2][r][c + o] = e11; hs[3][r][c + o] = e22; hs[4][r][c + o] = e12;
		}
	}
	__syncthreads();
	// columns: two vertically adjacent outputs per thread
	float acc[2][5];
	{
		float v[5][12];
#pragma unroll
		for (int q = 0; q < 5; q++)
#pragma unroll
			for (int k = 0; k < 12; k++) v[q][k] = hs[q][2 * ty2 + k][tx];
#pragma unroll
		for (int o = 0; o < 2; o++)
#pragma unroll
			for (int q = 0; q < 5; q++)
			{
				float t = 0;
#pragma unroll
				for (int k = 0; k < 11; k++) t = fmaf(c_win[k], v[q][k + o], t);
				acc[o][q] = t;
			}
	}
	const int px = bx + tx;
	float ssim_sum = 0.0f, l1_sum = 0.0f;
#pragma unroll
	for (int o = 0; o < 2; o++)
	{
		const int ly = 2 * ty2 + o, py = by + ly;
		if (px >= W || py >= H) continue;
		const float m1 = acc[o][0], m2 = acc[o][1], e11 = acc[o][2], e22 = acc[o][3], e12 = acc[o][4];
		const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
		const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
		const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
		const float a1 = 2.0f * mu12 + C1, a2 = 2.0f * s12 + C2, b1 = mu1_sq + mu2_sq + C1, b2 = s1 + s2 + C2;
		const float inv = 1.0f / (b1 * b2);
		const float ssim = a1 * a2 * inv;
		ssim_sum += ssim;
		l1_sum += fabsf(sx[ly + FR_LOSS_HALO][tx + FR_LOSS_HALO] - sy[ly + FR_LOSS_HALO][tx + FR_LOSS_HALO]);
		if (dmaps != nullptr)
		{
			const float df_dmu1 = 2.0f * m2 * a2 * inv - ssim * (2.0f * m1) / b1;
			const float df_ds1 = -ssim / b2;
			const float df_ds12 = 2.0f * a1 * inv;
			const size_t oo = ch * plane + (size_t)py * W + px, all = (size_t)C * plane;
			dmaps[oo] = df_dmu1 - 2.0f * m1 * df_ds1 - m2 * df_ds12;
			dmaps[all + oo] = df_ds1;
			dmaps[2 * all + oo] = df_ds12;
		}
	}
	// workgroup sums -> partials[block][0..1] = (sum |x - y|, sum ssim)
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { l1_sum += __shfl_xor(l1_sum, off); ssim_sum += __shfl_xor(ssim_sum, off); }
	if ((tid & 63) == 0) { s_red[0][tid >> 6] = l1_sum; s_red[1][tid >> 6] = ssim_sum; }
	__syncthreads();
	if (tid == 0)
	{
		const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
		partials[2 * b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
		partials[2 * b + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
	}
}

__global__ void __launch_bounds__(256) k_l1_ssim_bwd(int C, int H, int W, const float *__restrict__ x, const float *__restrict__ y,
	const float *__restrict__ dmaps, float w_l1, float w_ssim, const float *__restrict__ grad_scale, float *__restrict__ dL_dx)
{
	__shared__ float sm[3][FR_LOSS_SH][FR_LOSS_SW + 1];
	__shared__ float hs[3][FR_LOSS_SH][FR_LOSS_TW + 1];
	const int tid = threadIdx.x, tx = tid & 15, ty2 = tid >> 4;
	const int bx = blockIdx.x * FR_LOSS_TW, by = blockIdx.y * FR_LOSS_TH, ch = blockIdx.z;
	const size_t plane = (size_t)H * W, all = (size_t)C * plane;
	const float *M = dmaps + ch * plane;
	for (int i = tid; i < FR_LOSS_SH * FR_LOSS_SW; i += 256)
	{
		const int r = i / FR_LOSS_SW, c = i - r * FR_LOSS_SW;
		const int gy = by + r - FR_LOSS_HALO, gx = bx + c - FR_LOSS_HALO;
		const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
		const size_t o = (size_t)gy * W + gx;
		sm[0][r][c] = in ? M[o] : 0.0f;
		sm[1][r][c] = in ? M[all + o] : 0.0f;
		sm[2][r][c] = in ? M[2 * all + o] : 0.0f;
	}
	__syncthreads();
	for (int i = tid; i < FR_LOSS_SH * (FR_LOSS_TW / 2); i += 256)
	{
		const int r = i >> 3, c = (i & 7) * 2;
#pragma unroll
		for (int q = 0; q < 3; q++)
		{
			float v[12];
#pragma unroll
			for (int k = 0; k < 12; k++) v[k] = sm[q][r][c + k];
#pragma unroll
			for (int o = 0; o < 2; o++)
			{
				float t = 0;
#pragma unroll
				for (int k = 0; k < 11; k++) t = fmaf(c_win[k], v[k + o], t);
				hs[q][r][c + o] = t;
			}
		}
	}
	__syncthreads();
	float acc[2][3];
#pragma unroll
	for (int q = 0; q < 3; q++)
	{
		float v[12];
#pragma unroll
		for (int k = 0; k < 12; k++) v[k] = hs[q][2 * ty2 + k][tx];
#pragma unroll
		for (int o = 0; o < 2; o++)
		{
			float t = 0;
#pragma unroll
			for (int k = 0; k < 11; k++) t = fmaf(c_win[k], v[k + o], t);
			acc[o][q] = t;
		}
	}
	const int px = bx + tx;
	const float gs = grad_scale ? *grad_scale : 1.0f; // the loss' upstream gradient, still on the device
#pragma unroll
	for (int o = 0; o < 2; o++)
	{
		const int py = by + 2 * ty2 + o;
		if (px >= W || py >= H) continue;
		const size_t oo = ch * plane + (size_t)py * W + px;
		const float xv = x[oo], yv = y[oo];
		const float d = xv - yv;
		const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
		const float g = w_l1 * sgn + w_ssim * (acc[o][0] + 2.0f * xv * acc[o][1] + yv * acc[o][2]);
		dL_dx[oo] = grad_scale ? g * gs : g;
	}
}

int launch_l1_ssim_forward(int C, int H, int W, const float *x, const float *y, float *dmaps, float *partials, hipStream_t stream)
{
	const dim3 grid((W + FR_LOSS_TW - 1) / FR_LOSS_TW, (H + FR_LOSS_TH - 1) / FR_LOSS_TH, C);
	hipLaunchKernelGGL(k_l1_ssim_fwd, grid, dim3(256), 0, stream, C, H, W, x, y, dmaps, partials);
	return check_launch("l1_ssim_forward", stream, false);
}

// The per-tile partial sums -> (loss, l1, ssim), one workgroup: every thread adds its stride of the partials in double,
// the 1024 sums are added in a fixed tree (no atomics: the same bits on every run). Replaces a reduction, two divisions,
// two conversions and four scalar kernels of the host framework (~60 us of launches for ~100 KB of data).
__global__ void __launch_bounds__(1024) k_l1_ssim_finish(int nblocks, double n, const float *__restrict__ partials, float lam, float *__restrict__ out3)
{
	__shared__ double s0[1024], s1[1024];
	double a = 0.0, b = 0.0;
	for (int i = threadIdx.x; i < nblocks; i += 1024) { a += (double)partials[2 * i]; b += (double)partials[2 * i + 1]; }
	s0[threadIdx.x] = a; s1[threadIdx.x] = b;
	__syncthreads();
	for (int off = 512; off > 0; off >>= 1)
	{
		if ((int)threadIdx.x < off) { s0[threadIdx.x] += s0[threadIdx.x + off]; s1[threadIdx.x] += s1[threadIdx.x + off]; }
		__syncthreads();
	}
	if (threadIdx.x == 0)
	{
		const float l1 = (float)(s0[0] / n), ss = (float)(s1[0] / n);
		out3[0] = (1.0f - lam) * l1 + lam * (1.0f - ss);
		out3[1] = l1; out3[2] = ss;
	}
}

int launch_l1_ssim_finish(int nblocks, double n, const float *partials, float lam, float *out3, hipStream_t stream)
{
	hipLaunchKernelGGL(k_l1_ssim_finish, dim3(1), dim3(1024), 0, stream, nblocks, n, partials, lam, out3);
	return check_launch("l1_ssim_finish", stream, false);
}

int launch_l1_ssim_backward(int C, int H, int W, const float *x, const float *y, const float *dmaps, float w_l1, float w_ssim,
	const float *grad_scale, float *dL_dx, hipStream_t stream)
{
	const dim3 grid((W + FR_LOSS_TW - 1) / FR_LOSS_TW, (H + FR_LOSS_TH - 1) / FR_LOSS_TH, C);
	hipLaunchKernelGGL(k_l1_ssim_bwd, grid, dim3(256), 0, stream, C, H, W, x, y, dmaps, w_l1, w_ssim, grad_scale, dL_dx);
	return check_launch("l1_ssim_backward", stream, false);
}

} // namespace fr
