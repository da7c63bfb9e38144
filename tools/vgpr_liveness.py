"""VGPR liveness of one kernel from its ISA:  python tools/vgpr_liveness.py <object.o | lib.so> <substring of the demangled kernel name>
Disassembles the device code (llvm-objdump -d --line-numbers: build the object with -g for source lines), builds the control-flow
graph, runs a backward liveness analysis over the VGPRs (every definition kills: under divergence a masked definition does not, so the
true wave-level pressure is AT LEAST what is reported) and prints
  * the register count the analysis sees live at every basic block's entry and the maximum inside it,
  * at the point of maximum pressure: every live register with the source line that defined it and the source line of its next use,
    grouped by defining line -- i.e. which live ranges pin the kernel's register count."""
import collections, re, struct, subprocess, sys, tempfile, os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
NO_DEF = ("s_", "global_store", "flat_store", "scratch_store", "buffer_store", "ds_write", "ds_store", "v_cmp", "v_readfirstlane",
          "v_readlane", "v_nop", "global_load_lds", "buffer_load_lds", "ds_gws", "ds_nop", "global_atomic", "flat_atomic", "buffer_atomic",
          "global_wb", "global_inv", "buffer_wbl2", "buffer_inv")
DEF_AND_USE = ("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_dot4c", "v_writelane", "v_swap", "v_movreld", "v_accvgpr")


def device_elfs(path):
    b = open(path, "rb").read()
    i = b.find(b"\x7fELF", 0)
    while i != -1:
        if struct.unpack_from("<H", b, i + 18)[0] == 224:
            shoff = struct.unpack_from("<Q", b, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", b, i + 58)
            yield b[i:i + shoff + shentsize * shnum]
        i = b.find(b"\x7fELF", i + 1)


def vregs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def parse(text):
    """[(addr, opcode, defs, uses, source line)]"""
    ins, line = [], "?"
    for l in text.split("\n"):
        m = re.match(r"^; (\S+:\d+)", l)
        if m:
            line = m.group(1).split("/")[-1]
            continue
        m = re.match(r"\s+(\S+)(?:\s+(.*?))?\s+//\s*([0-9A-Fa-f]+):", l)
        if not m:
            continue
        op, rest, addr = m.group(1), m.group(2) or "", int(m.group(3), 16)
        toks = [t for t in re.split(r",\s*|\s+", rest) if t]
        regs = [vregs(t) for t in toks]
        defs, uses = [], []
        if op.startswith(NO_DEF) and not (op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")) and " glc" in l):
            for r in regs:
                uses += r
        else:
            first = True
            for r in regs:
                if not r:
                    first = False if first and False else first
                    continue
                if first:
                    defs += r
                    if op.startswith(DEF_AND_USE):
                        uses += r
                    first = False
                else:
                    uses += r
            # the destination is the FIRST operand only when that operand is a VGPR (v_mov_b32 v1, s2: yes; v_readlane s1, v2: handled above)
            if toks and not vregs(toks[0]):
                uses += defs
                defs = []
        ins.append((addr, op, defs, uses, line, rest))
    return ins


def main():
    path, pat = sys.argv[1], sys.argv[2]
    for elf in device_elfs(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(elf)
        txt = subprocess.run([OBJDUMP, "-d", "--demangle", "--line-numbers", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for m in re.finditer(r"^[0-9a-f]+ <(.*?)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S | re.M):
            if pat in m.group(1):
                report(m.group(1), parse(m.group(2)))
                return
    sys.exit("kernel not found")


def report(name, ins):
    n = len(ins)
    index = {a: i for i, (a, *_rest) in enumerate(ins)}
    succ = [[] for _ in range(n)]
    for i, (a, op, d, u, ln, rest) in enumerate(ins):
        if op == "s_endpgm":
            continue
        if op.startswith(("s_branch", "s_cbranch")):
            mm = re.search(r"(\d+)\s*$", rest)
            if mm:
                off = int(mm.group(1))
                off -= 65536 if off >= 32768 else 0
                t = a + 4 + 4 * off
                if t in index:
                    succ[i].append(index[t])
            if op.startswith("s_cbranch") and i + 1 < n:
                succ[i].append(i + 1)
        elif i + 1 < n:
            succ[i].append(i + 1)
    live_in = [frozenset()] * n
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            new = frozenset((out - set(ins[i][2])) | set(ins[i][3]))
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    peak = max(range(n), key=lambda i: len(live_in[i]))
    print(f"kernel {name[:120]}")
    print(f"{n} instructions; highest register index used: v{max([r for x in ins for r in x[2] + x[3]] or [0])}; "
          f"maximum of simultaneously live VGPRs seen by this analysis: {len(live_in[peak])} at instruction {peak} ({ins[peak][1]}, {ins[peak][4]})")
    # per source line: how many instructions, max live
    by_line = collections.OrderedDict()
    for i, x in enumerate(ins):
        e = by_line.setdefault(x[4], [0, 0])
        e[0] += 1
        e[1] = max(e[1], len(live_in[i]))
    print("\nlive VGPRs by source line (lines with >= 100 live somewhere; instruction count, maximum live):")
    for ln, (cnt, mx) in by_line.items():
        if mx >= 100:
            print(f"  {ln:34s} {cnt:5d} instructions, up to {mx} live")
    # who defined the registers live at the peak, and where are they used next
    last_def = {}
    def_line_at_peak = {}
    for i in range(peak + 1):
        for r in ins[i][2]:
            last_def[r] = ins[i][4]
    next_use = {}
    for i in range(peak, n):
        for r in ins[i][3]:
            next_use.setdefault(r, ins[i][4])
    groups = collections.defaultdict(list)
    for r in sorted(live_in[peak]):
        groups[last_def.get(r, "kernel argument / entry")].append(r)
    print(f"\nthe {len(live_in[peak])} registers live at the peak, by the source line of their (textually last) definition, with the line of a next use:")
    def source_text(ln):
        try:
            f, k = ln.rsplit(":", 1)
            for root in ("clive2_amd/csrc", "tests", "."):
                q = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), root, f)
                if os.path.exists(q):
                    return open(q).read().split("\n")[int(k) - 1].strip()[:110]
        except (ValueError, IndexError):
            pass
        return ""
    for ln, regs in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        uses = collections.Counter(next_use.get(r, "?") for r in regs)
        print(f"  {len(regs):3d} defined at {ln:28s} next used at " + ", ".join(f"{k} x{v}" for k, v in uses.most_common(3)))
        if len(regs) >= 3 and source_text(ln):
            print(f"        | {source_text(ln)}")


if __name__ == "__main__":
    main()
