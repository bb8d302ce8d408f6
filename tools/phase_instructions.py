"""Developer tool: static VALU instruction counts per phase of trace_kernel's main loop, from
the ISA that `make -C ray_tracing_amd/csrc asm` writes (built with -DRT_ASM_MARKS, which leaves a
`; RTMARK <phase>` comment at every phase boundary).  Instructions are attributed to the mark
that precedes them in layout order, which follows program order except for a few blocks the
compiler moves to the end.
    python tools/phase_instructions.py [kernel symbol substring]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.environ.get("RT_ISA") or os.path.join(ROOT, "ray_tracing_amd", "csrc", "asm", "rt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")


def main():
    sym = sys.argv[1] if len(sys.argv) > 1 else "trace_kernelILb0ELb0ELb0"
    lines = open(ISA).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    phase, order, counts = "prologue", ["prologue"], {"prologue": [0, 0, 0, 0]}
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"; RTMARK (\w+)", t)
        if m:
            phase = m.group(1)
            if phase not in counts:
                counts[phase] = [0, 0, 0, 0]
                order.append(phase)
            continue
        c = counts[phase]
        if t.startswith("v_"):
            c[0] += 1
            if t.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp", "v_sqrt", "v_rsq")):
                c[3] += 1
        elif t.startswith("s_"):
            c[1] += 1
        elif t.startswith(("ds_", "global_", "flat_", "buffer_", "scratch_")):
            c[2] += 1
    print("%-12s %6s %6s %6s %10s" % ("phase", "VALU", "SALU", "mem", "div/sqrt"))
    for ph in order:
        print("%-12s %6d %6d %6d %10d" % (ph, *counts[ph]))
    print("%-12s %6d" % ("total VALU", sum(c[0] for c in counts.values())))


if __name__ == "__main__":
    main()
