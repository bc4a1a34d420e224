#!/usr/bin/env python3
"""Human-readable table from a tools/pmc_summary.py JSON holding the MFMA / LDS counter passes.

    python tools/pmc_table.py profiles/r02_pmc_mfma_lds.json > profiles/r02_pmc_mfma_lds.summary.txt

GRBM_GUI_ACTIVE is summed over the 8 XCDs; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE / 8 * 1024 SIMDs);
LdsUtil = SQ_LDS_IDX_ACTIVE / (GUI_ACTIVE / 8 * 256 CUs); conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
MFMA flops = (SQ_INSTS_VALU_MFMA_MOPS_BF16 + ..._F16) * 512.
"""
import json
import sys


def main(path):
    d = json.load(open(path))
    print("# %s: per-dispatch means; MfmaUtil = MFMA busy cycles / (GUI_ACTIVE / 8 x 1024 SIMDs); LdsUtil = LDS_IDX_ACTIVE /" % path)
    print("# (GUI_ACTIVE / 8 x 256 CUs); conflicts = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE; MFMA GFLOP = MOPS_BF16 x 512")
    print("%-72s %5s %10s %8s %8s %9s %11s" % ("kernel", "calls", "cycles/XCD", "MfmaUtil", "LdsUtil", "conflicts", "MFMA GFLOP"))
    rows = []
    for k, v in d.items():
        act = v.get("GRBM_GUI_ACTIVE")
        if not act:
            continue
        cyc = act / 8.0
        rows.append((cyc * v.get("dispatches", 1), k, v, cyc))
    for _, k, v, cyc in sorted(rows, reverse=True)[:24]:
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES")
        la = v.get("SQ_LDS_IDX_ACTIVE")
        bc = v.get("SQ_LDS_BANK_CONFLICT")
        mo = v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16")
        if v.get("SQ_INSTS_VALU_MFMA_MOPS_F16"):        # (the fp16 rung of the similarity scan)
            mo = (mo or 0.0) + v["SQ_INSTS_VALU_MFMA_MOPS_F16"]
        f = lambda x: ("%7.1f%%" % x) if x is not None else "       -"
        print("%-72s %5d %10d %s %s %s %11s" % (k[:72], v.get("dispatches", 0), cyc,
                                              f(100.0 * mf / (cyc * 1024) if mf is not None else None),
                                              f(100.0 * la / (cyc * 256) if la is not None else None),
                                              (" %7.1f%%" % (100.0 * bc / la)) if (bc is not None and la) else "        -",
                                              ("%11.1f" % (mo * 512 / 1e9)) if mo is not None else "          -"))


if __name__ == "__main__":
    main(sys.argv[1])
