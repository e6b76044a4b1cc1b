#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection.csv files of one bench.py run into the per-kernel-group
JSON files kept under profiles/.

usage: summarize_pmc.py traffic <fetch.csv> <write.csv> <out.json> <passes> [workload description] [dtype] [batch size mode]
       summarize_pmc.py mfma <counters.csv> <out.json>
`passes` = workload passes in the profiled command (warm-up + timed + bench.py's own profiling pass).
Kernel groups are the names bench.py's in-library profiler uses."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LN_FOLD_PER_PASS = 88          # two LayerNorm folds per TransformerBlock, 44 blocks


def group(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    mm = re.match(r"_ZN3rst(?:12_GLOBAL__N_1)?(\d+)", n)          # an unmangled name rocprofv3 left alone
    if mm:
        start = mm.end()
        n = "rst::" + n[start:start + int(mm.group(1))] + "("
    m = re.match(r"rst::(\w+)<([^>]*)>", n)
    if not m:
        k = n.split("(")[0].replace("rst::", "")
        return {"attn_fold_kernel": "attn_fold", "attn_softmax_kernel": "attn_softmax", "ln_fold_weights_kernel": "ln_fold",
                "reduce_slabs_vec_kernel": "reduce_slabs", "reduce_slabs_kernel": "reduce_slabs",
                "gemm_f16s_prep_kernel": "gemm_f16s_prep", "conv3x3_f16s_prep_kernel": "conv3x3_f16s_prep",
                "dw_wgrad_pk_kernel": "pk_dw_wgrad", "gate_bwd_pk_kernel": "pk_gate_bwd", "gate_fwd_pk_kernel": "pk_gate"}.get(k, k)
    k, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if k in ("gemm_nn_res_kernel", "gemm_nn_dma_kernel"):
        return "gemm_nn_ln" if args[3] == "true" else "gemm_nn"
    if k == "gemm_f16s_kernel":
        return "gemm_f16s_ln" if args[2] == "true" else "gemm_f16s"
    if k == "pwdw_f16_kernel":
        return "pwdw_gate_f16" if args[3] == "true" else "pwdw_f16"
    if k == "conv3x3_f16s_kernel":
        return "conv3x3_f16s"
    if k == "gram_f16s_kernel":
        return "gram_f16s"
    if k == "conv3x3_pk_kernel":
        return "pk_conv3x3"
    if k == "wgrad_pk_kernel":
        return "pk_wgrad"
    if k == "gemm_pk_kernel":
        return "pk_gemm_ln" if args[2] == "true" else "pk_gemm"
    if k == "dw_pk_kernel":
        return "pk_dw_gate" if args[0] == "true" else "pk_dw"
    if k == "gram_pk_kernel":
        return "pk_gram"
    if k in ("pk_pack_kernel", "pk_unpack_kernel"):
        return k.replace("_kernel", "")
    if k == "ln_bwd_pk_kernel":
        return "pk_ln_bwd"
    if k == "ln_fwd_pk_kernel":
        return "pk_ln"
    if k == "pwdw_ring_kernel":
        return "pk_pwdw_gate" if args[1] == "true" else "pk_pwdw"
    if k == "pwdw_kernel":
        return "pwdw_gate" if args[2] == "true" else "pwdw"
    if k == "pwdw_ws_kernel":
        f16s = len(args) > 3 and args[3] == "true"
        return ("pwdw_gate" if args[2] == "true" else "pwdw") + ("_f16s" if f16s else "")
    return {"gemm_nt_kernel": "gemm_nt_splitk", "dwconv_fwd_kernel": "dwconv3x3", "dwconv_gate_fwd_kernel": "dwconv3x3_gate",
            "pixel_shuffle2_kernel": "pixel_shuffle2+pixel_unshuffle2", "conv3x3_fewout_kernel": "conv3x3_fewout",
            "gemm_nn_kernel": "gemm_nn"}.get(k, k)


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name = {}
    for r in csv.DictReader(open(path)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    return per, name


def passes_of(name):
    """Workload passes in the profiled command: two LayerNorm folds per block per fp32 pass, one Gram launch
    per block per bf16 pass."""
    folds = sum(1 for n in name.values() if "ln_fold" in n)
    if folds:
        return folds / LN_FOLD_PER_PASS
    return sum(1 for n in name.values() if "gram_pk_kernel" in n) / (LN_FOLD_PER_PASS / 2)


def traffic(fetch_csv, write_csv, out, passes, workload="bench.py default (Restormer dim=48 fp32 inference, 16x3x256x256)", dtype="f32",
            batch=16, size=256, mode="infer"):
    import bench
    f, fn = load(fetch_csv)
    w, wn = load(write_csv)
    passes = float(passes)
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for d, v in f.items():
        agg[group(fn[d])][0] += 1
        agg[group(fn[d])][1] += v["FETCH_SIZE"]
    for d, v in w.items():
        agg[group(wn[d])][2] += v["WRITE_SIZE"]
    kernels = {}
    for g, (n, fe, wr) in sorted(agg.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2])):
        lp = n / passes
        fb, wb = 2 * fe / passes * 1024, wr / passes * 1024
        kernels[g] = {"launches_per_step": lp, "fetch_bytes_per_step": round(fb), "write_bytes_per_step": round(wb),
                      "bytes_per_launch": round((fb + wb) / lp)}
    json.dump({"workload": workload,
               "dtype": dtype,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --steps 1 --warmup 1 --no-side-runs; "
                         "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); KB -> bytes x1024; "
                         f"{passes:g} passes in the profiled command",
               "config": {"batch": int(batch), "size": int(size), "mode": mode},
               "kernel_sources_sha": bench.kernel_sources_hash(),
               "total_bytes_per_step": round(sum(k["fetch_bytes_per_step"] + k["write_bytes_per_step"] for k in kernels.values())),
               "kernels": kernels}, open(out, "w"), indent=1)


def mfma(counters_csv, out):
    per, name = load(counters_csv)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for d, v in per.items():
        g = group(name[d])
        cnt[g] += 1
        for k, x in v.items():
            agg[g][k] += x
    kernels = {}
    for g, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        if busy == 0:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        kernels[g] = {"launches": cnt[g], "mfma_instructions": int(v["SQ_INSTS_MFMA"]), "mfma_busy_cycles": int(busy),
                      "gpu_cycles": int(cyc), "mfma_utilisation": round(busy / (cyc * 1024), 4)}
    json.dump({"workload": "bench.py --steps 1 --warmup 1 (Restormer dim=48 fp32 inference, 16x3x256x256)",
               "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE; "
                         "GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GPU cycles = GRBM_GUI_ACTIVE / 8; matrix-pipe "
                         "utilisation = MFMA busy cycles / (GPU cycles x 1024 SIMDs); a v_mfma_f32_16x16x4_f32 holds the pipe 32 cycles",
               "kernels": kernels}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:11])
    else:
        mfma(sys.argv[2], sys.argv[3])
