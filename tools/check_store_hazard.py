"""Developer tool: scan hipcc's assembly for 128-bit buffer stores with an SGPR soffset whose data registers a vector instruction
writes within the next two wait states.  LLVM's hazard recognizer does not pad that case (it assumes the SGPR offset's extra issue
cycle covers it); on gfx950 the store was measured to pick up the new value in some lanes.
usage: hipcc -S --cuda-device-only ... -o k.s file.hip; python tools/check_store_hazard.py k.s"""
import re, sys



def scan(path):
    """-> list of (kernel symbol, line number, store, overwriting instruction)"""
    lines = open(path).read().split("\n")
    return _scan_lines(lines)


st = re.compile(r"\s*buffer_store_dwordx[34] v\[(\d+):(\d+)\], \w+, s\[\d+:\d+\], (s\d+|\d+)")
dst = re.compile(r"\s*(v_\w+) (v\[(\d+):(\d+)\]|v(\d+))")


def _scan_lines(lines):
    kern, found = None, []
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.rstrip().split(":")[0].endswith("E"):
            kern = l.split(":")[0]
        m = st.match(l)
        if not m or not m.group(3).startswith("s"):
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        ws, j = 0, i + 1
        while ws < 2 and j < len(lines):
            t = lines[j].strip()
            j += 1
            if not t or t.startswith(";") or t.startswith("."):
                if t.startswith(".LBB"):     # (a branch target: the fall-through path is scanned, a jump into the block is not)
                    break
                continue
            n = re.match(r"s_nop (\d+)", t)
            if n:
                ws += int(n.group(1)) + 1
                continue
            d = dst.match(lines[j - 1])
            # (an MFMA writes its result at the END of its passes, far behind the store's register read: not a hazard)
            if d and not d.group(1).startswith("v_cmp") and not d.group(1).startswith("v_mfma"):
                a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
                if a <= hi and b >= lo:
                    found.append((kern, i + 1, l.strip(), t))
            ws += 1
    return found


if __name__ == "__main__":
    hits = scan(sys.argv[1])
    for kern, ln, store, instr in hits:
        print(f"{kern}: line {ln}: {store}  <-  {instr}")
    print("hazards:", len(hits))
