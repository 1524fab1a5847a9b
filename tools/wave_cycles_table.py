"""Where the waves' cycles go, per kernel (round 5): from the `pmc_wave_cycles*.txt` of tools/profile_round.sh (rocprofv3 --pmc
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES).
    parked  = SQ_WAIT_ANY      / SQ_WAVE_CYCLES   a wave sits at s_waitcnt / a barrier (memory latency, hand-overs)
    stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   a wave has an instruction and the pipe does not take it (fp32 MFMA back to back:
                                                  k_clock_probe, a pure MFMA stream, reads 76 %)
    issuing = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (of which VALU incl. MFMA issue, LDS, vector memory)
The three are disjoint and add up to ~100 % (MI355X_MICROARCH.md, rocprofv3 PMC slots).
usage: python tools/wave_cycles_table.py gpurun_out/<tag>/pmc_wave_cycles.txt > profiles/<tag>_wave_cycles.md"""
import sys


def main():
    rows, cur = {}, None
    for line in open(sys.argv[1]):
        if line.strip() and not line.startswith(" "):
            cur = line.strip()
            rows[cur] = {}
        elif line.strip():
            p = line.split()
            rows[cur][p[0]] = (int(p[2]), float(p[4]))
    print("| kernel | launches profiled | wave quad-cycles per launch | parked % | stalled at issue % | issuing % | VALU (incl. MFMA) % | LDS % | vector memory % |")
    print("|---|---|---|---|---|---|---|---|---|")
    key = lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0))[1] * kv[1].get("SQ_WAVE_CYCLES", (1, 0))[0]
    for k, c in sorted(rows.items(), key=key):
        w = c.get("SQ_WAVE_CYCLES", (0, 0.0))[1]
        if w <= 0 or k.startswith(("k_copy_from_host", "k_fill", "__amd")):
            continue
        g = lambda n: 100.0 * c.get(n, (0, 0.0))[1] / w
        name = k.replace("void ", "").split("(")[0]
        print(f"| `{name[:70]}` | {c['SQ_WAVE_CYCLES'][0]} | {w:.0f} | {g('SQ_WAIT_ANY'):.1f} | {g('SQ_WAIT_INST_ANY'):.1f} | "
              f"{g('SQ_ACTIVE_INST_ANY'):.1f} | {g('SQ_ACTIVE_INST_VALU'):.1f} | {g('SQ_ACTIVE_INST_LDS'):.1f} | {g('SQ_ACTIVE_INST_VMEM'):.1f} |")


if __name__ == "__main__":
    main()
