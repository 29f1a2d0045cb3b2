#!/usr/bin/env python3
"""Build profiles/rNN_pmc_sq_<workload>.json from the three rocprofv3 --pmc SQ_* passes of profiles/run_profiles.sh <round> <workload> sq.

usage: make_sq_json.py <sq1_results.db> <sq2_results.db> <sq3_results.db> <workload> <batch> <horizon> <round>

Per kernel (template variants that run side by side in every step share a key, like make_traffic_json.py): the issue counters summed
over the variants per step, per workgroup figures, and the derived fractions bench.py quotes:
  valu_lane_slots_per_launch = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 64 + SQ_INSTS_MFMA x 256      lane slots issued on the vector + matrix pipes
                               (SQ_INSTS_VALU counts the matrix instructions too; an FP64 v_mfma_f64_4x4x4_4b is 256 multiply-adds = 16 cycles x 16 lanes;
                               the 16x16x4 form is 1024 -- kernels that use it say so in `mfma_macs_per_inst`)
  mfma_util                  = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)                matrix pipe busy over the four SIMDs of the busy CUs
  valu_busy                  = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES-normalised (see code)   vector pipe busy
  lds_bank_conflict_frac     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import json
import re
import sqlite3
import sys

MFMA_MACS = {"ocp_riccati_backward_reg": None, "parnmpc_kkt_inverse_wave": None}      # mixed 16x16x4 / 4x4x4 kernels: lane slots from busy cycles instead


def key_of(name):
    m = re.search(r"idocp_dev::(\w+?)_kernel(<[^>]*?(false|true)?[^>]*>)?", name)
    if not m:
        return None
    key = m.group(1)
    if "condense" in key and re.search(r"_kernel<[^,>]+,\s*true", name):
        key += "_residual"
    elif key.startswith("un_linearize") and re.search(r"_kernel<\d+,\s*1\b", name):
        key += "_residual"
    return key


def collect(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select k.name, p.name, count(distinct e.event_id), sum(e.value) from rocpd_pmc_event e "
        "join rocpd_info_pmc p on e.pmc_id = p.id join kernels k on k.id = e.event_id group by k.name, p.name").fetchall()
    out = {}
    for kname, cname, launches, total in rows:
        key = key_of(kname)
        if key is None:
            continue
        cur = out.setdefault(key, {}).get(cname, (0, 0.0))
        out[key][cname] = (max(cur[0], launches), cur[1] + total)
    return {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in out.items()}


def main():
    dbs, (workload, batch, horizon, rnd) = sys.argv[1:4], sys.argv[4:8]
    merged = {}
    for path in dbs:
        for k, d in collect(path).items():
            merged.setdefault(k, {}).update(d)
    kernels = {}
    for k, c in sorted(merged.items()):
        g = lambda n: c.get(n, 0.0)
        wgs = None
        rec = {"counters_per_launch": {n: c[n] for n in sorted(c)}}
        mf, valu = g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU")
        mfma_slots = 16.0 * g("SQ_VALU_MFMA_BUSY_CYCLES")                     # busy cycles x 16 lanes: right for every FP64 MFMA shape
        rec["valu_lane_slots_per_launch"] = (valu - mf) * 64.0 + mfma_slots
        rec["mfma_util"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES")) if g("SQ_BUSY_CU_CYCLES") else None
        rec["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None
        # vector pipe busy: SQ_ACTIVE_INST_VALU is in quad-cycles per wave-instruction; over the SIMD cycles of the busy CUs (pass 3's SQ_BUSY_CU_CYCLES
        # is per CU: x 4 SIMDs)
        rec["valu_busy"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (4.0 * g("SQ_BUSY_CU_CYCLES")) if g("SQ_BUSY_CU_CYCLES") else None
        if g("SQ_WAVES"):
            rec["per_wave"] = {"valu": valu / g("SQ_WAVES"), "salu": g("SQ_INSTS_SALU") / g("SQ_WAVES"), "lds": g("SQ_INSTS_LDS") / g("SQ_WAVES"),
                               "mfma": mf / g("SQ_WAVES"), "vmem_rd": g("SQ_INSTS_VMEM_RD") / g("SQ_WAVES"), "vmem_wr": g("SQ_INSTS_VMEM_WR") / g("SQ_WAVES")}
        kernels[k] = rec
    json.dump({"round": int(rnd), "workload": workload, "batch": int(batch), "horizon": int(horizon),
               "command": "rocprofv3 --kernel-trace --pmc SQ_* (three separate passes, profiles/run_profiles.sh %s %s sq) -- python3 bench.py --workload %s --steps 3 --warmup 1 "
                          "--no-cpu-baseline --no-latency" % (rnd, workload, workload),
               "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
