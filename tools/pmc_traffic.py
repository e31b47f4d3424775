#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (``--pmc FETCH_SIZE`` and ``--pmc WRITE_SIZE``, each with --kernel-trace only) into
HBM bytes per launch per instrumented site, as MI355X_MICROARCH.md prescribes for gfx950:

    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        # FETCH_SIZE counts 128-byte units in KB-of-64B on gfx950

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic_latest.json ["source note"]
"""
import csv
import glob
import json
import sys
from collections import defaultdict

SITE_OF = (("lstm_step_mfma_pair", "gemm_lstm_rec"), ("score_mfma_kernel<21, 8, 2", "score_fused"), ("score_mfma_kernel<6", "score_fused"),
           ("score_pairs_video_kernel", "score_pairs"), ("score_pairs_exact_kernel", "score_pairs"), ("segment_pool_norm", "pool"), ("topk_merge", "topk_merge"),
           ("gemm_nt_mfma<true, 2>", "gemm_vis_seg"))


def per_kernel(folder, counter):
    tot, cnt = defaultdict(float), defaultdict(set)
    for path in glob.glob(f"{folder}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] += float(row["Counter_Value"])
            cnt[name].add(row["Dispatch_Id"])
    return {k: tot[k] / len(cnt[k]) for k in tot}, {k: len(v) for k, v in cnt.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    out, detail = {}, {}
    for name in fetch:
        b = (2.0 * fetch[name] + write.get(name, 0.0)) * 1024.0
        detail[name[:120]] = {"bytes_per_launch": b, "launches": nf[name]}
        for pat, site in SITE_OF:
            if pat in name and (site not in out or b > out[site]):
                out[site] = b
    if len(sys.argv) > 4:
        out["_source"] = sys.argv[4]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    json.dump(detail, open(sys.argv[3].replace(".json", "_detail.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
