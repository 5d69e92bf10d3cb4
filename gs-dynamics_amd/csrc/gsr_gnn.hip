// gsr_gnn.hip -- the two kernels of the particle dynamics' propagation network (DynamicsPredictor: /root/reference/src/gnn/model.py:70-246, run
// per rollout step by /root/reference/src/gnn/dynamics_module.py:53-172) that are NOT matrix products.  The products stay with the GEMM
// library; the algebra that shrinks them is in gsdyn/dynamics.py (_propagate_split): the relation propagator's
//   cat(relation_encode, effect[receiver], effect[sender]) @ [W1 | W2 | W3]^T  =  relation_encode @ W1^T + b   (once per call)
//                                                                               + (effect @ W2^T)[receiver] + (effect @ W3^T)[sender]
// -- products on the N nodes instead of the E relations.  (Round 4 also built the WHOLE network as one persistent launch on the f32 MFMA
// with device-wide barriers: correct to 1e-6 and 2x slower than this path, 379 vs ~185 us per call -- profiles/r04_gnn_one_launch.txt has
// the analysis; the kernel left the tree in round 5.)
#include "gsr_common.h"

// ---------------------------------------------------------------- pieces for the GEMM-library path (the default)
// The same algebra with the products left to the library: what remains between them is (a) the relation inputs and (b) the aggregation
//   agg[i] = sum over the relations e of receiver i, in list order, of relu(rew1[e] + a2[i] + a3[sender_e])
// -- (b) replaces, per propagation step, two row gathers, a concatenation, an E x 3H x H product, a zero fill and an index_add.
namespace gsr_gnn {
__global__ __launch_bounds__(256) void aggregate_kernel(int N, int n_sum, int H, const float* __restrict__ rew1, const float* __restrict__ a23,
                                                        const long long* __restrict__ send, const long long* __restrict__ row_start,
                                                        float* __restrict__ agg, const float* __restrict__ res_a, const float* __restrict__ res_b,
                                                        float* __restrict__ res_out) {
  const int o = blockIdx.x * 256 + threadIdx.x, q = H >> 2;
  if (o >= N * q) return;
  const int i = o / q, k4 = (o - i * q) << 2;
  if (res_out) {   // the particle propagator's addend of this step, res_a + res_b (= particle_encode @ Wp1^T + b, and the effect as the residual):
    const float4 x = *reinterpret_cast<const float4*>(res_a + (size_t)i * H + k4), y = *reinterpret_cast<const float4*>(res_b + (size_t)i * H + k4);
    *reinterpret_cast<float4*>(res_out + (size_t)i * H + k4) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);   // one launch less per step
  }
  const float4 a2 = *reinterpret_cast<const float4*>(a23 + (size_t)i * 2 * H + k4);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int e1 = i < n_sum ? (int)row_start[i + 1] : 0;      // rows >= n_sum (the padding's dummy row, which collects every dummy relation) get zeros
  for (int e = i < n_sum ? (int)row_start[i] : 0; e < e1; ++e) {
    const float4 r1 = *reinterpret_cast<const float4*>(rew1 + (size_t)e * H + k4);
    const float4 a3 = *reinterpret_cast<const float4*>(a23 + (size_t)send[e] * 2 * H + H + k4);
    sum.x += fmaxf((r1.x + a2.x) + a3.x, 0.f); sum.y += fmaxf((r1.y + a2.y) + a3.y, 0.f);
    sum.z += fmaxf((r1.z + a2.z) + a3.z, 0.f); sum.w += fmaxf((r1.w + a2.w) + a3.w, 0.f);
  }
  *reinterpret_cast<float4*>(agg + (size_t)i * H + k4) = sum;
}
// rel_inputs[e] = (attrs[recv], attrs[send], sum |instance[recv] - instance[send]|, state[recv] - state[send]); nodes = [attrs | instance | state]
__global__ __launch_bounds__(256) void rel_inputs_kernel(int E, int A, int G, int S, const float* __restrict__ nodes,
                                                         const long long* __restrict__ recv, const long long* __restrict__ send,
                                                         float* __restrict__ out) {
  const int Dr = 2 * A + 1 + S, F = A + G + S, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= E * Dr) return;
  const int e = o / Dr, c = o - e * Dr;
  const float* __restrict__ xr = nodes + (size_t)recv[e] * F;
  const float* __restrict__ xs = nodes + (size_t)send[e] * F;
  float v;
  if (c < A) v = xr[c];
  else if (c < 2 * A) v = xs[c - A];
  else if (c == 2 * A) { v = 0.f; for (int k = 0; k < G; ++k) v += fabsf(xr[A + k] - xs[A + k]); }
  else v = xr[A + G + (c - 2 * A - 1)] - xs[A + G + (c - 2 * A - 1)];
  out[o] = v;
}
}  // namespace gsr_gnn

int gsr_launch_gnn_aggregate(int N, int n_sum, int H, const float* rew1, const float* a23, const long long* send, const long long* row_start, float* agg,
                             hipStream_t st, const float* res_a, const float* res_b, float* res_out) {
  { GSR_PROF("gnn_aggregate", st);
    hipLaunchKernelGGL(gsr_gnn::aggregate_kernel, dim3((N * (H >> 2) + 255) / 256), dim3(256), 0, st, N, n_sum, H, rew1, a23, send, row_start, agg, res_a, res_b,
                       res_out); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
int gsr_launch_gnn_rel_inputs(int E, int A, int G, int S, const float* nodes, const long long* recv, const long long* send, float* out,
                              hipStream_t st) {
  { GSR_PROF("gnn_rel_inputs", st);
    hipLaunchKernelGGL(gsr_gnn::rel_inputs_kernel, dim3((E * (2 * A + 1 + S) + 255) / 256), dim3(256), 0, st, E, A, G, S, nodes, recv, send, out); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

