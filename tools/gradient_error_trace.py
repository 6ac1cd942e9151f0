"""Where does each arithmetic's gradient error enter?  For the cases of tools/arithmetic_error_report.py: rel-L2 error against the
float64 oracle of the gradient ARRIVING at every traced activation (backward order), for the HIP default arithmetic, the HIP kernels
with fp32 operands and torch fp32 on the CPU, over several (weights seed, data seed) draws -- one draw is one sample of an
ill-conditioned problem (Large: the top-k pooling's score gradient is a cancelling sum over the channels), so a claim about an
arithmetic needs the spread.  Test tooling (imports oracle/).
    python tools/gradient_error_trace.py [--case large] [--draws 3] [--no-trace] [--exact-gemm]
Run against another build:  python tools/run_with_lib.py <lib.so> tools/gradient_error_trace.py ..."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import arithmetic_error_report as R  # noqa: E402


def exact_gemms():
    """--exact-gemm: every tile GEMM of the HIP legs computed by torch in float64 and rounded once (a diagnostic stand-in, this tool only):
    what is left is the error of everything that is NOT a dense contraction."""
    import torch
    from dgdm_histopath_lab_amd import ops

    def nt(a, w, bias=None, out=None, accumulate=False, math="fp32"):
        r = a.double() @ w.double().t()
        if bias is not None:
            r = r + bias.double()
        if out is not None:
            out.copy_((r + out.double()).float() if accumulate else r.float())
            return out
        return r.float()

    def nt_split(a, w0, w1, bias=None, math="bf16x3"):
        return nt(a, torch.cat([w0, w1], dim=1), bias)

    def nn(a, w, out=None, accumulate=False, math="fp32"):
        r = a.double() @ w.double()
        if out is not None:
            out.copy_((r + out.double()).float() if accumulate else r.float())
            return out
        return r.float()

    def tn(dy, x, with_bias, math="fp32", split=None, out=None, may_defer=False):
        dW = (dy.double().t() @ x.double()).float()
        db = dy.double().sum(0).float() if with_bias else None
        if split is not None:
            return (dW[:, :split].contiguous(), dW[:, split:].contiguous()), db
        if out is not None:
            out.copy_(dW)
            return out, db
        return dW, db

    ops.gemm_nt_raw, ops.gemm_nt_split_raw, ops.gemm_nn_raw, ops.gemm_tn_raw = nt, nt_split, nn, tn


def main():
    a = sys.argv[1:]
    if "--exact-gemm" in a:
        exact_gemms()
        print("# --exact-gemm: tile GEMMs of both HIP legs replaced by float64 torch products (diagnostic)")
    case = a[a.index("--case") + 1] if "--case" in a else "large"
    draws = int(a[a.index("--draws") + 1]) if "--draws" in a else 3
    want_trace = "--no-trace" not in a
    summ = []
    for d in range(draws):
        res = R.run_case(case, seed=3 + d, data_seed=d, trace_grads=want_trace)
        s = R.summary(res["rows"])
        summ.append(s)
        print("# %s draw %d (weights seed %d, data seed %d): parameter gradients max | median   default %.2e | %.2e   HIP fp32 %.2e | %.2e   "
              "torch fp32 %.2e | %.2e" % (case, d, 3 + d, d, *s[0], *s[1], *s[2]))
        if res["trace_rows"]:
            print("%-20s %-10s %-10s %-10s %-8s| %-10s %-10s %-10s %s" % ("gradient arriving at", "default", "HIP fp32", "torch fp32", "HIP/torch",
                                                                          "value: def", "HIP fp32", "torch fp32", "HIP/torch"))
            for k, e0, e1, e2, v0, v1, v2 in res["trace_rows"]:
                print("%-20s %-10.2e %-10.2e %-10.2e %-8.2f| %-10.2e %-10.2e %-10.2e %.2f" % (k, e0, e1, e2, e1 / max(e2, 1e-30), v0, v1, v2,
                                                                                          v1 / max(v2, 1e-30)))
        sys.stdout.flush()
    print("# over %d draws, median of the per-draw (max | median): default %.2e | %.2e   HIP fp32 %.2e | %.2e   torch fp32 %.2e | %.2e" % (
        draws, *(statistics.median(s[i][j] for s in summ) for i in range(3) for j in range(2))))
    print("# ratio to torch fp32 per draw (median column): default %s   HIP fp32 %s" % (
        " ".join("%.2f" % (s[0][1] / s[2][1]) for s in summ), " ".join("%.2f" % (s[1][1] / s[2][1]) for s in summ)))


if __name__ == "__main__":
    main()
