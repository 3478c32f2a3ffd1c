"""The VI step of the 122 880-row shard of C3 with the scale hierarchy on the side stream (SPMF_VI_OVERLAP=1), for a
rocprofv3 --kernel-trace: do the side stream's kernels run BESIDE the column pass, or take turns with it?
usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/vi_overlap_trace.py
       python3 tools/vi_overlap_trace.py --analyze <dir>"""
import contextlib
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last complete step: from the last-but-one sample_noise of the D group backwards
    names = [r["Kernel_Name"].split("(")[0].replace("void spmf::", "")[:40] for r in rows]
    ends = [i for i, n in enumerate(names) if "surrogate_bwd_adam" in n]
    lo, hi = ends[-3] + 1, ends[-2] + 1
    t0 = int(rows[lo]["Start_Timestamp"])
    print(f"{'kernel':42s} {'queue':>6s} {'start_us':>9s} {'end_us':>9s} {'dur_us':>8s}")
    for r, n in zip(rows[lo:hi], names[lo:hi]):
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{n:42s} {r.get('Queue_Id', '?'):>6s} {s / 1e3:9.2f} {e / 1e3:9.2f} {(e - s) / 1e3:8.2f}")
    sys.exit(0)

os.environ["SPMF_VI_OVERLAP"] = "1"
import torch  # noqa: E402

sys.path.insert(0, ".")
from spmf_amd import PoissonFactorization, synth, vi  # noqa: E402
from spmf_amd.sparse import balanced_panel_rows  # noqa: E402

D, K, rows = 20_000, 32, 122_880
dev = torch.device("cuda", 0)
sc = synth.linear_structure(rows, D, 0.005, dev, panel_rows=balanced_panel_rows(rows, K))
with contextlib.redirect_stdout(sys.stderr):
    m = PoissonFactorization(latent_dim=K, feature_dim=D, u_tau_scale=1.0 / (rows * D) ** 0.5, device=dev)
colsum = torch.zeros(D, dtype=torch.float64, device=dev)
colnnz = torch.zeros_like(colsum)
sc.compute_stats(m._handle(), colsum, colnnz)
cm = colsum / colnnz
m.eta_i = torch.where(cm > 1, cm, torch.ones_like(cm)).reshape(1, D)
m.xi_u_global = float(torch.nansum(cm))
opt = vi.AdamHIP(m, m.surrogate_distribution.trainable_variables, 1e-3)
opt.init_state(3.0)
run = vi.StepRunner(m, opt, rows, 1, use_graph=True, seed=5)
for _ in range(30):
    run.step({"counts": sc})
torch.cuda.synchronize()
print("done", opt.read_state()[11])
