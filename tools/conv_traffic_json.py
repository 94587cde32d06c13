"""Assemble profiles/rNN_dominant_conv_traffic.json from the counter averages of tools/final_evidence.sh (gpurun_out/final/conv_pmc.txt: separate
rocprofv3 --pmc passes over tools/one_conv.py, FETCH_SIZE / WRITE_SIZE in KiB per dispatch).  FETCH_SIZE is doubled (gfx950 correction,
/opt/skills/guides/MI355X_MICROARCH.md HBM section); WRITE_SIZE is taken as reported.  usage: conv_traffic_json.py <conv_pmc.txt> <out.json> [crops]"""
import json
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    crops = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    sec = {}
    cur = None
    for line in txt.splitlines():
        m = re.match(r"== (\S+)", line)
        if m:
            cur = sec.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s+(\S+)\s+([0-9.e+-]+)\s+\(n=(\d+)\)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    fetch_kb, write_kb = sec["conv_FETCH_SIZE"]["FETCH_SIZE"], sec["conv_WRITE_SIZE"]["WRITE_SIZE"]
    rd, wr = 2.0 * fetch_kb * 1024, write_kb * 1024
    alg = crops * 128 * 128 * 512 * 2 * 2 + 512 * 9 * 512 * 2        # input + output fp16 + weights
    busy, gui = sec["conv_mfma"]["SQ_VALU_MFMA_BUSY_CYCLES"], sec["conv_mfma"]["GRBM_GUI_ACTIVE"]
    out = {"kernel": f"3x3 conv 512->512 @128x128, {crops} crops per launch (tools/one_conv.py -1 5 {crops} 128 512 512: the cost model's tile)",
           "crops_per_launch": crops, "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE), averages over the dispatches of "
                   "tools/one_conv.py; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported",
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
           "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui,
           "mfma_busy_share_at_actual_clock": busy / (gui / 8.0 * 1024.0)}
    if "conv128_mfma" in sec:
        b2, g2 = sec["conv128_mfma"]["SQ_VALU_MFMA_BUSY_CYCLES"], sec["conv128_mfma"]["GRBM_GUI_ACTIVE"]
        out["conv128_level"] = {"kernel": "3x3 conv 128->128 @512x512, 16 crops (tools/one_conv.py -1 5 16 512 128 128)", "SQ_VALU_MFMA_BUSY_CYCLES": b2,
                                "GRBM_GUI_ACTIVE": g2, "busy_share_at_actual_clock": b2 / (g2 / 8.0 * 1024.0)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
