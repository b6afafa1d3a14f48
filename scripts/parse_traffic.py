#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE): average bytes per launch, per kernel and -- for the 256-tile GEMM --
per SHAPE of the BEATs layer loop.

    python scripts/parse_traffic.py <dir with <tag>_pmc_fetch/ and <tag>_pmc_write/> <tag>

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half the bytes of wide
(16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.  The counters sit between the L2s
and the fabric: Infinity Cache hits are counted too.

Per-shape rows: kernel names do not tell QKV from fc1 (same instantiation), but a forward launches its GEMMs in a fixed order after
its fbank_kernel: patch_embed, post_extract_proj, then (qkv, out_proj, fc1, fc2) per layer.  Rows are walked in dispatch order and the
256-tile launches of every f16 forward with the benchmark's grid are labelled by position.
"""
import collections
import csv
import glob
import json
import os
import sys

SHORT = ("gemm256p_kernel", "gemm_nt_kernel", "attention3_kernel", "attention2_kernel", "attention_tail_kernel", "attention_kernel", "layernorm_half_kernel",
         "layernorm_pool_kernel", "layernorm_kernel", "ln_rowstats_kernel", "posconv_kernel", "fbank_kernel", "mean_pool_kernel")
LAYER = ("qkv", "out_proj", "fc1", "fc2")


def short_name(k: str):
    for s in SHORT:
        if s in k:
            return s
    return None


def is_bf16(k: str) -> bool:
    return "DF16b" in k or "__bf16" in k or "bf16" in k


def main() -> int:
    root, tag = sys.argv[1], sys.argv[2]
    out = collections.defaultdict(lambda: {"launches": 0})
    shapes = collections.defaultdict(dict)
    for name, sub, scale in (("fetch", f"{tag}_pmc_fetch", 2.0 * 1024), ("write", f"{tag}_pmc_write", 1.0 * 1024)):
        files = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
        per = collections.defaultdict(list)
        per_shape = collections.defaultdict(list)
        for f in files:
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
            fwd = None          # the 256-tile GEMM launches of the current f16 forward, None outside one

            def close(fwd):
                # a forward whose launch count is not 2 + 4 L (another epilogue mix, a hooked forward, a smaller batch in between) is skipped
                if fwd and len(fwd) > 2 and (len(fwd) - 2) % 4 == 0:
                    for pos, (grid, v) in enumerate(fwd):
                        label = ("patch_embed", "post_extract_proj")[pos] if pos < 2 else LAYER[(pos - 2) % 4]
                        if grid == 256 * 512:          # full-size launches only (the benchmark's batch)
                            per_shape[label].append(v)
            for r in rows:
                k = r["Kernel_Name"]
                s = short_name(k)
                if s is None:
                    continue
                v = float(r["Counter_Value"]) * scale
                if not is_bf16(k):
                    per[s].append(v)
                if s == "fbank_kernel":
                    close(fwd)
                    fwd = None if is_bf16(k) else []
                elif s == "gemm256p_kernel" and fwd is not None:
                    if is_bf16(k):
                        fwd = None
                    else:
                        fwd.append((int(r["Grid_Size"]), v))
            close(fwd)
        for k, v in per.items():
            out[k][f"{name}_bytes_per_launch"] = sum(v) / len(v)
            out[k]["launches"] = len(v)
        for k, v in per_shape.items():
            shapes[k][f"{name}_bytes_per_launch"] = sum(v) / len(v)
            shapes[k]["launches"] = len(v)
    res = {k: dict(v, hbm_bytes_per_launch=v.get("fetch_bytes_per_launch", 0) + v.get("write_bytes_per_launch", 0)) for k, v in out.items()}
    res["gemm256p_kernel_by_shape"] = {k: dict(v, hbm_bytes_per_launch=v.get("fetch_bytes_per_launch", 0) + v.get("write_bytes_per_launch", 0))
                                       for k, v in shapes.items()}
    lay = [res["gemm256p_kernel_by_shape"].get(k) for k in LAYER]
    if all(lay):       # the dominant kernel as the bench line defines it: the four layer GEMMs
        res["gemm256p_kernel_layer_loop"] = {"hbm_bytes_per_launch": sum(x["hbm_bytes_per_launch"] for x in lay) / 4.0,
                                            "note": "mean over qkv / out_proj / fc1 / fc2 (12 launches each per forward)"}
    path = os.path.join(root, f"{tag}_traffic.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))
    return 0


if __name__ == "__main__":
    sys.exit(main())
