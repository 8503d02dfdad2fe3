#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of bench.py -> where the wall time of ONE training step goes: the last step in the
trace is cut at the stem kernels; for the main queue, the time between consecutive conv3x3 / marker kernels is
attributed to phases (forward per stage, backward per stage, optimizer).  usage: step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r["Queue_Id"]) for r in rows))
stems = [i for i, e in enumerate(ev) if e[2].startswith("stem_conv_kernel")]
a, b = stems[-2], stems[-1]
step = ev[a:b]
t0, t1 = step[0][0], ev[b][0]
print(f"step wall {1e-3 * (t1 - t0):.1f} us, {len(step)} kernels")
# phase boundaries: forward ends at soft_loss / head kernel; optimizer starts at sgd_kernel
def first(pred, lst=step):
    return next(e for e in lst if pred(e[2]))
loss_k = first(lambda n: "soft_loss_kernel" in n or "head_soft_loss_kernel" in n)
sgd_k = first(lambda n: n.startswith("sgd_kernel"))
print(f"forward  {1e-3 * (loss_k[0] - t0):8.1f} us")
print(f"backward {1e-3 * (sgd_k[0] - loss_k[0]):8.1f} us")
print(f"optimizer + next-step prep {1e-3 * (t1 - sgd_k[0]):8.1f} us")
# busy / idle of the union of all queues
iv = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU idle (no kernel on any queue) {1e-3 * ((t1 - t0) - busy):8.1f} us")
# forward: time by kernel family between t0 and loss
def fam(n):
    for k in ("conv3x3_pp_kernel", "conv_igemm_dma_kernel", "conv_wgrad_ks_kernel", "conv_wgrad_pp_kernel", "conv_wgrad_dma_kernel", "bn_apply_kernel", "bn_fold_partials",
              "bn_bwd_reduce_cus", "bn_bwd_apply_cus", "bn_bwd_finalize", "bn_bwd_reduce_kernel", "bn_bwd_apply_kernel", "weight_", "sgd", "Fill"):
        if k in n:
            return k
    return "other"
import collections
for name, lo, hi in (("forward", t0, loss_k[0]), ("backward", loss_k[0], sgd_k[0])):
    d = collections.defaultdict(float)
    gaps = 0.0
    main = [e for e in step if lo <= e[0] < hi and e[3] == step[0][3]]
    for e in main:
        d[fam(e[2])] += e[1] - e[0]
    for x, y in zip(main, main[1:]):
        gaps += max(0, y[0] - x[1])
    print(f"-- {name}: main-queue kernel time by family (us), launch gaps {1e-3 * gaps:.1f} us over {len(main)} kernels")
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]):
        print(f"   {k:28s} {1e-3 * v:8.1f}")
# backward per unit: split at the dense dgrad launches
