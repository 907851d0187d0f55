#!/usr/bin/env python3
"""Per-kernel summary (JSON) of the two SQ counter passes tools/pmc_sq_headline.sh / pmc_sq_config.sh write to sq.txt:
    python tools/sq_summary.py gpurun_out/<dir>/sq.txt "<what was profiled>" > profiles/rNN_sq_counters.json
Averages per launch; SQ_WAVE_CYCLES and the ACTIVE / WAIT counters are in units of four cycles."""
import ast
import json
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    d = {}
    for i in range(0, len(txt) - 1, 2):
        name = txt[i].strip()
        m = re.match(r"\s*(\{.*\}) launches (\d+)", txt[i + 1]) if name else None
        if m:
            d.setdefault(name, {}).update(ast.literal_eval(m.group(1)))
    out = {"source": "rocprofv3 --pmc, two passes (tools/pmc_sq_headline.sh or pmc_sq_config.sh), " + (sys.argv[2] if len(sys.argv) > 2 else ""),
           "units": "averages per launch; cycles in units of four (wave_cycles_per_wave = 4 x the counter)", "kernels": {}}
    for k, v in d.items():
        w, wc = v.get("SQ_WAVES", 0), v.get("SQ_WAVE_CYCLES", 0)
        if not w or not wc:
            continue
        out["kernels"][k] = {
            "waves": w, "wave_cycles_per_wave": round(4 * wc / w),
            "valu_active_frac": round(v["SQ_ACTIVE_INST_VALU"] / wc, 3), "lds_active_frac": round(v["SQ_ACTIVE_INST_LDS"] / wc, 3),
            "wait_any_frac": round(v["SQ_WAIT_ANY"] / wc, 3), "wait_inst_any_frac": round(v["SQ_WAIT_INST_ANY"] / wc, 3),
            "insts_per_wave": {"valu": round(v["SQ_INSTS_VALU"] / w), "salu": round(v["SQ_INSTS_SALU"] / w), "lds": round(v["SQ_INSTS_LDS"] / w),
                               "vmem": round(v["SQ_INSTS_VMEM"] / w)},
            "lds_bank_conflict_frac": round(v["SQ_LDS_BANK_CONFLICT"] / max(1, v["SQ_LDS_IDX_ACTIVE"]), 3)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
