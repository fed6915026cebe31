// libgantts_hip.so -- launcher of the fused discriminator stack (dstack_f32.hip.h): one launch per discriminator pass for the
// hidden layers above the first one + the head (+ the backward-data chain of the generator step's adversarial term).
#include "engine_internal.hip.h"
#include "dstack_f32.hip.h"

using namespace gt;

bool dstack_hidden_ok(int hidden_dim) { return hidden_dim == 128 || hidden_dim == 256; }

template <int HD>
static int launch_dstack_t(const DStackArgs& a, hipStream_t s) {
  const size_t lds = dstack_lds_bytes<HD>();
  CHK(ensure_dyn_lds((const void*)dstack_kernel<HD>, lds));
  const int grid = cdiv(a.rows, DS_R);
  if (grid <= 0) return GT_OK;
  GemmProfiler::Rec rec;
  const bool prof = g_prof.wants(7);
  if (prof) {       // bench.py's per-kernel table: kind 7 = the fused discriminator stack (slot 14)
    const double both = a.mode == DSTACK_G_ADV && a.want_grad ? 2.0 : 1.0;
    rec.kind = 7; rec.bn = 64; rec.am = -1;
    rec.flops = 2.0 * a.rows * HD * HD * (a.L - 1) * both + (both > 1.0 ? 2.0 * a.rows * HD * a.Da : 0.0) + 2.0 * a.rows * HD;
    // algorithmic bytes: H0 once, the weights once, the D step's stashes (L - 1 activations + the seed gradient) once
    rec.bytes = 4.0 * ((double)a.rows * HD + (double)(a.L - 1) * HD * HD + (a.mode == DSTACK_D_STEP ? (double)a.L * a.rows * HD : (double)a.rows * a.Da));
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL((dstack_kernel<HD>), dim3(grid), dim3(DS_THREADS), lds, s, a);
  LAUNCH_CHECK();
  if (prof) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}

// panels of the pass = workgroups = entries of a.hp (and rows of a.dw_partial)
int dstack_panels(long rows) { return cdiv(rows, DS_R); }

int launch_dstack(const DStackArgs& a, int hidden_dim, hipStream_t s) {
  if (a.L < 1 || a.L > DS_MAXL) return fail(GT_ERR_INVALID, "fused discriminator stack: 1 .. %d hidden layers", DS_MAXL);
  if (a.mode == DSTACK_G_ADV && a.want_grad && (a.Da < 1 || a.Da > 64)) return fail(GT_ERR_INVALID, "fused discriminator stack: 1 .. 64 adversarial columns");
  if (hidden_dim == 256) return launch_dstack_t<256>(a, s);
  if (hidden_dim == 128) return launch_dstack_t<128>(a, s);
  return fail(GT_ERR_INVALID, "fused discriminator stack: hidden_dim 128 or 256");
}
