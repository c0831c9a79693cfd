/* fftconv_amd.h -- C ABI of libfftconv_amd.so (MI355X / gfx950 FFT convolution).
 *
 * The reference (klae01/fft-conv-pytorch) has no FFI: its boundary is the Python
 * signature fft_conv(signal, kernel, bias, stride, padding, dilation, groups,
 * padding_mode) at fft_conv_pytorch/functional.py:19-28.  This ABI is what a
 * native backend for that function binds to; each entry point cites the piece
 * of the reference it replaces.  Plain pointers and sizes only -- no torch
 * types.  All device buffers are owned by the caller; the library owns only the
 * plan (device twiddle tables + launch geometry).  Every launch is asynchronous
 * on the caller's stream; no entry point on the hot path allocates or
 * synchronises.  Status: 0 = OK, non-zero = error, text via fc_last_error().
 */
#ifndef FFTCONV_AMD_H
#define FFTCONV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FC_ABI_VERSION 6

enum fc_status {
  FC_OK = 0,
  FC_ERR_INVALID = 1,      /* bad argument / shape (ValueError on the Python side) */
  FC_ERR_UNSUPPORTED = 2,  /* valid but outside what this build handles */
  FC_ERR_HIP = 3           /* a HIP runtime call failed */
};

enum fc_pad_mode { FC_PAD_CONSTANT = 0, FC_PAD_REFLECT = 1, FC_PAD_REPLICATE = 2, FC_PAD_CIRCULAR = 3 };
enum fc_dtype {
  FC_F32 = 0,  /* the FFT kernels; every float* below is float */
  FC_F64 = 1   /* float64 tensors: x, weight, w_hat, bias and y are double (pass them through the float* / void*
                  parameters); a direct time-domain kernel computes the same function in float64 (the reference is
                  dtype-agnostic); fc_wgrad1d and the profiling hook are float32-only */
};

/* Problem descriptor: the arguments of functional.py:19-28 after to_ntuple
 * (utils.py:4-20) has been applied on the host side.  Axis order is the tensor
 * order: spatial[0] is the slowest spatial axis. */
typedef struct fc_desc {
  int32_t ndim;          /* 1, 2 or 3 spatial axes */
  int32_t dtype;         /* fc_dtype */
  int64_t batch;
  int64_t in_channels;
  int64_t out_channels;
  int64_t groups;
  int64_t spatial[3];    /* input extent per axis (unpadded) */
  int64_t kernel[3];     /* kernel taps per axis */
  int64_t stride[3];
  int64_t padding[3];
  int64_t dilation[3];
  int32_t padding_mode;  /* fc_pad_mode; "zeros" of nn.Conv == constant (nn.py:12) */
  int32_t has_bias;
  int32_t tile_hint;     /* 0 = auto; otherwise force the 1-D FFT tile length */
  int32_t transposed;    /* 0: fft_conv (functional.py:19-89); 1: fft_conv_transpose (functional.py:92-176):
                            kernel is (Cin, Cout/groups, *k), stride spreads the input, padding crops the
                            output, output_padding extends it; padding_mode must be constant */
  int64_t output_padding[3];   /* transposed only */
} fc_desc;

typedef struct fc_plan fc_plan;

/* ABI version of the loaded library (FC_ABI_VERSION). */
int fc_version(void);

/* Thread-local text of the last error returned on this thread. */
const char* fc_last_error(void);

/* Validate the descriptor, choose tiles and build device twiddle tables.
 * Host-side arithmetic of functional.py:44-47,66 (a1, a4). May allocate. */
int fc_plan_create(const fc_desc* desc, fc_plan** out_plan);
void fc_plan_destroy(fc_plan* plan);

/* Output extent per spatial axis: floor((S + 2p - d(k-1) - 1)/stride) + 1,
 * the slice arithmetic of functional.py:76-82 (a9); for a transposed plan
 * (S-1)*stride - 2p + d(k-1) + output_padding + 1 (functional.py:144-154). */
int fc_output_shape(const fc_plan* plan, int64_t out_spatial[3]);

/* Bytes the caller must provide for the transformed kernel / the scratch area (N-d plans and the many-channel
 * 1-D pipeline use a scratch area, in fc_transform_kernel AND fc_forward; 0 for the fused 1-D kernels). */
size_t fc_kernel_spectrum_bytes(const fc_plan* plan);
size_t fc_workspace_bytes(const fc_plan* plan);

/* FFT tile length chosen for the last (fused) axis, for reporting. */
int fc_plan_tile(const fc_plan* plan);

/* Everything the byte layout of the kernel spectrum depends on besides the descriptor itself:
 * {tile, dilation phases, kernel segments, taps per segment, depthwise blocks, regrouped small groups,
 * wide-input kernel (2 = the many-channel pipeline: bin-major complex matrices for its per-bin GEMM), batch items
 * per workgroup}.  Two plans of equal descriptor (up to the batch size),
 * equal fc_kernel_spectrum_bytes() and equal layout words accept each other's fc_transform_kernel()
 * output -- what a multi-GPU caller checks before broadcasting one rank's spectrum (the planner looks at
 * the local batch size).  No counterpart in the reference (it re-transforms the kernel on every call,
 * functional.py:71). */
int fc_plan_layout(const fc_plan* plan, int32_t layout[8]);

/* Kernel transform: dilation scatter + zero pad + real FFT + conjugate
 * (functional.py:49-57 and :71; rows a2, a6).  weight is (Cout, Cin/groups, *k)
 * contiguous fp32 on the device; w_hat receives fc_kernel_spectrum_bytes(). */
int fc_transform_kernel(const fc_plan* plan, const float* weight, void* w_hat, void* workspace,
                        void* hip_stream);

/* Forward convolution (functional.py:60-87; rows a3, a5, a7-a10): padding,
 * forward real FFT, per-bin grouped channel contraction, inverse FFT, valid
 * window, stride, bias -- x is (B, Cin, *spatial), y is (B, Cout, *out) fp32
 * contiguous on the device.  bias may be NULL. */
int fc_forward(const fc_plan* plan, const float* x, const void* w_hat, const float* bias, float* y,
               void* workspace, void* hip_stream);

/* Weight gradient of the 1-D convolution described by `desc` (the FORWARD descriptor):
 *   dW[o][i][k] = sum_b sum_t dY[b][o][t] * pad(x)[b][i][t + k*dilation]
 * i.e. what autograd derives from functional.py:60-87 for `kernel` (pinned by the reference's
 * tests/test_functional.py:111-117).  Covered: ndim 1, stride 1, <= 64 channels per group on both
 * sides, any kernel length and padding mode (long kernels run in segments of taps; depthwise shapes
 * with a multiple of 8 channels take a per-channel variant).  Cross-spectra are accumulated over the batch
 * and the row on chip; the result comes in `slices` partial tensors that the caller sums:
 *   partial is (slices, Cout, Cin/groups, K) fp32, fully written by the call.
 * fc_wgrad1d_slices returns the slice count for the current device, 0 when the shape is not covered
 * (the caller then differentiates through fc_forward plans instead); it also builds the device tables
 * the launch needs, so fc_wgrad1d itself never allocates or copies (call fc_wgrad1d_slices first, on the
 * same device -- the caller needs its answer to size `partial` anyway). */
int fc_wgrad1d_slices(const fc_desc* desc);
int fc_wgrad1d(const fc_desc* desc, const float* x, const float* dy, float* partial, int slices, void* hip_stream);
/* The same launch with the bias gradient folded in (ABI 5): db[o] = sum over batch and row of dY[b][o][t] is bin 0 of the
 * gradient spectra the kernel forms anyway, so the separate reduction over dY (tests/test_functional.py:114-117 pins db)
 * costs nothing.  Slice s of the result starts at partial + s*slice_stride floats (0 = densely packed) and at
 * db_partial + s*slice_stride: one buffer of `slices` rows [dW | db] with db_partial = partial + Cout*Cin/groups*K is
 * summed by ONE reduction.  db_partial may be NULL.  Not available for depthwise plans (fc_wgrad1d_db_supported = 0). */
int fc_wgrad1d_db_supported(const fc_desc* desc);
int fc_wgrad1d_db(const fc_desc* desc, const float* x, const float* dy, float* partial, float* db_partial,
                  long long slice_stride, int slices, void* hip_stream);

/* Weight gradient of a 2-D / 3-D convolution (ABI 6; row N1: the reference's dW comes out of autograd through its
 * rfftn / einsum / irfftn graph, tests/test_functional.py:62-117 pin it for ndim 1-3, stride 1-2, groups 1-3):
 *   dW[(g,o)][i][k] = sum_b sum_t dY[b][(g,o)][t] * Xpad[b][(g,i)][t*stride + k*dilation]
 * is the convolution of x with batch and channels exchanged against dY (stride and dilation exchanged too), of which
 * the first kernel[i] lags per axis are kept.  fc_wgrad_nd_plan_create takes the descriptor OF THE CONVOLUTION and
 * returns the plan of that gradient (destroy with fc_plan_destroy; fc_kernel_spectrum_bytes / fc_workspace_bytes size the
 * two scratch buffers).  fc_wgrad_nd transforms dY (B, Cout, *Lout), runs x (B, Cin, *S) against it and writes
 * dw (Cout, Cin/groups, *k) completely -- both tensors are read and dW is written in these layouts, no transposed
 * copies, no partial results.  Asynchronous on hip_stream; allocates nothing. */
int fc_wgrad_nd_plan_create(const fc_desc* conv_desc, fc_plan** out_plan);
int fc_wgrad_nd(const fc_plan* plan, const float* x, const float* dy, float* dw, void* spectrum, void* workspace,
                void* hip_stream);

/* Profiling variant of fc_forward (not part of the drop-in surface; the plan stays immutable):
 * `stamps` is a device buffer of 16 * fc_debug_grid(plan) uint64 in which lane 0 of the waves of the
 * fused kernels stores the 100 MHz wall clock at its phase boundaries (batch-sharing 1-D kernel: one
 * record of 16 stamps per wave, 16 wave slots per work item).  The stamped launch drains its loads at
 * two extra points: read shares, not totals. */
int fc_forward_stamped(const fc_plan* plan, const float* x, const void* w_hat, const float* bias, float* y,
                       void* workspace, void* hip_stream, void* stamps);
long long fc_debug_grid(const fc_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* FFTCONV_AMD_H */
