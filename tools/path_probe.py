#!/usr/bin/env python3
"""Whole-step time of the read operations over one configuration of tools/profile_op.py, once per read path:

    python3 tools/path_probe.py [cfg5sin] [slots,vstream,auto] [filter,masks,decode]

Prints one line per (path, operation): milliseconds per step over PROBE_STEPS (200) steps after PROBE_WARM (300)
warm-up steps.  With the probe build (CVEC_LIB_PATH=.../libcvec_hip_probes.so), CVEC_RANGES_PER_WAVE (0 = one
fixed range per wave) and CVEC_WORK_GROUPS set how the section kernel hands out its slot ranges."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import compressed_vec_amd as cv  # noqa: E402
from compressed_vec_amd import _native as N  # noqa: E402
import profile_op  # noqa: E402

PATHS = {"slots": N.PATH_SLOTS, "auto": N.PATH_AUTO, "vstream": N.PATH_VSTREAM}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5sin"
    paths = (sys.argv[2] if len(sys.argv) > 2 else "slots").split(",")
    ops = (sys.argv[3] if len(sys.argv) > 3 else "filter").split(",")
    warm, steps = int(os.environ.get("PROBE_WARM", "300")), int(os.environ.get("PROBE_STEPS", "200"))
    assert cv.lib().cvec_init(0) == 0
    tname, vals, eo, pred, add_const = profile_op.build(name)
    batch = cv.encode_device(tname, vals, eo)
    out = torch.empty(batch.n_sections * 256, dtype=vals.dtype, device="cuda") if "decode" in ops else None
    masks = torch.empty(batch.n_sections * 32, dtype=torch.uint8, device="cuda") if "masks" in ops else None
    res = batch.result_buffer()
    p = cv.make_pred(batch.etype, pred)
    raw = vals.numel() * vals.element_size()
    want = None
    for path in paths:
        assert cv.lib().cvec_set_read_path(PATHS[path]) == 0
        for op in ops:
            def step():
                if op == "filter":
                    batch.filter_count_async(p, result=res)
                elif op == "masks":
                    batch.filter_count_async(p, result=res, masks=masks)
                else:
                    batch.decode_async(out, add_const=add_const, result=res)
            for _ in range(warm):
                step()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(steps):
                step()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / steps
            r = res.cpu().numpy()
            if op == "filter":
                want = int(r[0]) if want is None else want
                assert int(r[0]) == want and int(r[1]) == 0, (r, want)
            print("%s %-7s %-6s %.4f ms/step  %.0f GB/s  result %s" % (name, path, op, ms, raw / ms / 1e6, r[:3].tolist()), flush=True)
    cv.lib().cvec_set_read_path(N.PATH_AUTO)


if __name__ == "__main__":
    main()
