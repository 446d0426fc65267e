#!/usr/bin/env python3
"""Per-kernel memory-side traffic from the L2's request counters BY REQUEST SIZE (round 4).

Round 3 priced a kernel's FETCH_SIZE with one factor (x2.0, calibrated on wide coalesced
streams).  FETCH_SIZE is RDREQ x 64 B with the 32-byte requests taken out, i.e. it tallies the
128-byte requests of a coalesced stream at 64 B (MI355X_MICROARCH.md, HBM) — and a kernel
whose reads are gathers issues 64-byte requests, which the x2 then doubles (the round-3 FM
forward "moved 6.6 TB/s").  The L2 counts its requests to the fabric by size, so no factor is
needed:

    read bytes  = 32 x RDREQ_32B + 128 x RDREQ_128B + 64 x (RDREQ - RDREQ_32B - RDREQ_128B)
    write bytes = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B)

  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum \\
      --kernel-trace --output-format csv -d out/rd -- <cmd> --pmc-calibrate
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ... -d out/wr -- <cmd> --pmc-calibrate
  python tools/pmc_traffic2.py out/rd out/wr > profiles/rNN/pmc_traffic_<workload>.json

The same run streams and gathers known byte counts (xf_calib_stream: 1 GiB, beyond the 256 MiB
Infinity Cache; coalesced reads / writes at 4, 8, 16 bytes per lane; gathers of one 4-byte word
per 128-byte line, of 32-byte records, of 64- and 256-byte rows, every unit once): the
`calibration` block reports measured / true bytes for each pattern — the check that the formula
holds for the access patterns the kernels use (a 4-byte gather's "true" bytes are given both as
the bytes used and as the lines it touches)."""
import csv
import glob
import json
import os
import re
import sys

GIB = 1 << 30
CALIB = {   # kernel-name fragment -> (label, useful bytes, bytes of the lines / sectors touched)
    "k_calib_read<unsigned int>": ("stream read 4 B/lane", GIB, GIB),
    "k_calib_read<unsigned long>": ("stream read 8 B/lane", GIB, GIB),
    "k_calib_read<HIP_vector_type<unsigned int, 4u>": ("stream read 16 B/lane", GIB, GIB),
    "k_calib_write<unsigned int>": ("stream write 4 B/lane", GIB, GIB),
    "k_calib_write<unsigned long>": ("stream write 8 B/lane", GIB, GIB),
    "k_calib_write<HIP_vector_type<unsigned int, 4u>": ("stream write 16 B/lane", GIB, GIB),
    "k_calib_gather<4, 128>": ("gather 4 B of every 128-B line", GIB // 32, GIB),
    "k_calib_gather<32, 32>": ("gather 32-B records", GIB, GIB),
    "k_calib_gather<64, 64>": ("gather 64-B rows", GIB, GIB),
    "k_calib_gather<256, 256>": ("gather 256-B rows", GIB, GIB),
}


def kname(full):
    m = re.search(r"(k_[a-z0-9_]+)(<.*?>)?\(", full)
    if not m:
        return full.split("(")[0].strip()
    return m.group(1) + (m.group(2) or "")


def load(d):
    """{kernel: [{counter: value, 'us': duration}, ...]} in launch order"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        by_dispatch = {}
        for r in csv.DictReader(open(f)):
            e = by_dispatch.setdefault(r["Dispatch_Id"], {
                "k": kname(r["Kernel_Name"]),
                "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for _, e in sorted(by_dispatch.items(), key=lambda kv: int(kv[0])):
            out.setdefault(e["k"], []).append(e)
    return out


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else 0.0


def rd_bytes(e):
    r, r32, r128 = e.get("TCC_EA0_RDREQ_sum", 0), e.get("TCC_EA0_RDREQ_32B_sum", 0), \
        e.get("TCC_EA0_RDREQ_128B_sum", 0)
    return 32 * r32 + 128 * r128 + 64 * (r - r32 - r128)


def wr_bytes(e):
    w, w64 = e.get("TCC_EA0_WRREQ_sum", 0), e.get("TCC_EA0_WRREQ_64B_sum", 0)
    return 64 * w64 + 32 * (w - w64)


def main():
    rd, wr = load(sys.argv[1]), load(sys.argv[2])
    workload = None
    for d in sys.argv[1:3]:
        try:
            workload = json.loads(open(d.rstrip("/") + ".json").read().strip().splitlines()[-1])[
                "config"]["workload"]
            break
        except (OSError, ValueError, KeyError, IndexError):
            pass
    out = {"workload": workload, "unit": "bytes per launch (median over the later half of a "
           "kernel's launches)", "method": "L2 -> fabric requests by size: read = 32 x RDREQ_32B "
           "+ 128 x RDREQ_128B + 64 x the rest; write = 64 x WRREQ_64B + 32 x the rest "
           "(tools/pmc_traffic2.py); no per-pattern factor", "calibration": {}, "kernels": {}}
    for frag, (label, useful, lines) in CALIB.items():
        src = wr if "write" in frag else rd
        ks = [k for k in src if k.startswith(frag)]
        if not ks:
            continue
        es = src[ks[0]]
        b = median([(wr_bytes if "write" in frag else rd_bytes)(e) for e in es])
        us = median([e["us"] for e in es])
        out["calibration"][label] = {
            "useful_bytes": useful, "bytes_of_the_lines_touched": lines, "measured_bytes": b,
            "measured_over_useful": b / useful, "measured_over_lines": b / lines,
            "median_us": us, "useful_gbs": useful / us / 1e3, "measured_gbs": b / us / 1e3}
    for k in sorted(set(rd) & set(wr)):
        if k.startswith("k_calib") or not k.startswith("k_"):
            continue
        re_, we_ = rd[k][len(rd[k]) // 2:], wr[k][len(wr[k]) // 2:]
        fb, wb = median([rd_bytes(e) for e in re_]), median([wr_bytes(e) for e in we_])
        us = median([e["us"] for e in re_])
        r = median([e.get("TCC_EA0_RDREQ_sum", 0) for e in re_]) or 1.0
        out["kernels"][k] = {
            "launches": len(rd[k]), "read": fb, "write": wb, "traffic": fb + wb,
            "read_requests": {"n": r,
                              "share_32B": median([e.get("TCC_EA0_RDREQ_32B_sum", 0) for e in re_]) / r,
                              "share_128B": median([e.get("TCC_EA0_RDREQ_128B_sum", 0) for e in re_]) / r},
            "median_us_under_pmc": us, "gbs_under_pmc": (fb + wb) / us / 1e3 if us else None}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
