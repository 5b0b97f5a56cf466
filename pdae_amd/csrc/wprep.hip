// Grouped weight preparation: the fragment-ordered split copies of EVERY trainable convolution weight of a plan (forward and data-gradient
// forms, 3x3 / fused 1x1 skip / 1x1 layouts) in ONE launch at the head of the step, instead of one few-microsecond launch in front of each
// convolution (125 + 9 per FFHQ-128 training step: 0.8 ms of launches for 1.3 GB/s-trivial work; per denoising step of a sampling loop: ~60).
// The weights change once per optimizer step, so once per plan run is exactly as often as needed.
// Block b -> job by binary search in the prefix table `first_block`; each thread prepares one 8-element fragment slot (conv3x3p.h).
#include "common.h"
#include "igemm.h"
#include "conv3x3p.h"

template <int NS> __device__ __forceinline__ void wprep_job_slot(const WprepJob& j, size_t i) {
  if (j.T > 0) {
    if (i < (size_t)(j.C >> 5) * 2 * j.T * j.NT * 64) {
      if (j.transposed & PDAE_WPREP_FORM_X) wprepx_slot<NS>(j.w, j.Nout, j.C, j.NT, j.transposed & 1, j.wscale, j.T, j.wp, i);
      else wprep3_slot<NS>(j.w, j.Nout, j.C, j.NT, j.transposed, j.wscale, j.T, j.wp, i);
    }
  } else {
    if (i < (size_t)(j.C >> 4) * j.NT * 64) wprep1_slot<NS>(j.w, j.Nout, j.C, j.NT, j.transposed, j.wscale, j.wp, i);
  }
}

__global__ void __launch_bounds__(256) wprep_group_kernel(const WprepJob* __restrict__ jobs, const int* __restrict__ first_block, int njobs) {
  const int b = blockIdx.x;
  int lo = 0, hi = njobs - 1;                       // last job whose first block is <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const WprepJob j = jobs[lo];
  const size_t i = (size_t)(b - first_block[lo]) * 256 + threadIdx.x;
  switch (j.ns) {
    case 1: wprep_job_slot<1>(j, i); break;
    case 2: wprep_job_slot<2>(j, i); break;
    case 4: wprep_job_slot<4>(j, i); break;
    default: wprep_job_slot<3>(j, i); break;
  }
}

int wprep_group_launch(const WprepJob* jobs_dev, const int* first_block_dev, int njobs, int total_blocks, hipStream_t s) {
  if (njobs <= 0 || total_blocks <= 0) return PDAE_OK;
  hipLaunchKernelGGL(wprep_group_kernel, dim3(total_blocks), dim3(256), 0, s, jobs_dev, first_block_dev, njobs);
  return pdae_launch_status("wprep_group");
}
