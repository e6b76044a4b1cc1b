"""What this MI355X delivers on a streaming copy / read and on bare MFMA loops (rst_peak_copy / rst_peak_mfma),
next to the paper peaks that bench.py's roofline fractions use.  Writes one JSON object (stdout, and --out FILE).

    python tools/peaks.py --out profiles/r3_peaks.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PAPER = {"hbm_GBps": 8000.0, "mfma_f32_TFLOPs": 157.3, "mfma_bf16_TFLOPs": 2516.6, "mfma_f16_TFLOPs": 2516.6}


def _time(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def measure():
    from superresolution_amd import _lib
    from superresolution_amd._lib import check
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    out = {"device": torch.cuda.get_device_name(0), "paper": PAPER}
    nbytes = 2 << 30                                  # 2 GiB per buffer: far beyond the 256 MiB last-level cache
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda").fill_(1)
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    t = _time(lambda: check(lib.rst_peak_copy(a.data_ptr(), b.data_ptr(), nbytes, 0, st), "rst_peak_copy"), 20)
    out["copy_GBps"] = 2 * nbytes / t / 1e9
    t = _time(lambda: check(lib.rst_peak_copy(a.data_ptr(), b.data_ptr(), nbytes, 1, st), "rst_peak_copy"), 20)
    out["read_GBps"] = nbytes / t / 1e9
    t = _time(lambda: b.copy_(a), 20)
    out["torch_copy_GBps"] = 2 * nbytes / t / 1e9
    del a, b
    # the mixes the network's memory-bound kernels have, by torch's elementwise kernels on 800 MB operands:
    # c = a + b (two reads, one write: a residual product), a += 1 (read and write of the same lines), a fill
    n = 200 * 1024 * 1024
    x, y, z = (torch.rand(n, device="cuda") for _ in range(3))
    t = _time(lambda: torch.add(x, y, out=z), 20)
    out["triad_GBps"] = 12 * n / t / 1e9
    t = _time(lambda: x.add_(1.0), 20)
    out["inplace_GBps"] = 8 * n / t / 1e9
    t = _time(lambda: z.fill_(1.0), 20)
    out["fill_GBps"] = 4 * n / t / 1e9
    out["stream_GBps"] = max(out["copy_GBps"], out["torch_copy_GBps"], out["triad_GBps"])     # best read + write stream
    del x, y, z
    iters = 4000
    buf = torch.empty(256 * 8 * 256, dtype=torch.float32, device="cuda")
    for kind, name, k in ((0, "mfma_f32_16x16x4", 4), (1, "mfma_bf16_16x16x32", 32), (2, "mfma_f16_16x16x32", 32)):
        sweep = {}
        for wps in (1, 2, 4, 8):                      # blocks of 4 waves per CU = waves per SIMD
            blocks = 256 * wps
            t = _time(lambda: check(lib.rst_peak_mfma(kind, blocks, iters, buf.data_ptr(), st), "rst_peak_mfma"), 5)
            sweep[str(wps)] = round(blocks * 4 * iters * 8 * 2.0 * 16 * 16 * k / t / 1e12, 1)
        out[name + "_TFLOPs"] = max(sweep.values())
        out[name + "_by_waves_per_simd"] = sweep
    try:
        out["git_head"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    args = ap.parse_args()
    res = measure()
    text = json.dumps(res, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
