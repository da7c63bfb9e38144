"""Basic-block summary of one kernel of a built library:  python tools/isa_blocks.py <lib.so> <substring of the demangled kernel name> [--dump]
Per block: instruction range, VALU / SALU / memory instruction counts, the memory opcodes, the terminator and its target."""
import re, struct, subprocess, sys, tempfile, os

def device_elfs(path):
    b = open(path, 'rb').read()
    i = b.find(b'\x7fELF', 1)
    while i != -1:
        if struct.unpack_from('<H', b, i + 18)[0] == 224:
            shoff = struct.unpack_from('<Q', b, i + 40)[0]
            shentsize, shnum = struct.unpack_from('<HH', b, i + 58)
            yield b[i:i + shoff + shentsize * shnum]
        i = b.find(b'\x7fELF', i + 1)

def main():
    lib, pat = sys.argv[1], sys.argv[2]
    dump = '--dump' in sys.argv
    for elf in device_elfs(lib):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(elf)
        txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '--demangle', f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for m in re.finditer(r'^[0-9a-f]+ <(.*?)>:\n(.*?)(?=^\n|\Z)', txt, re.S | re.M):
            if pat in m.group(1):
                report(m.group(1), m.group(2), dump)

def report(name, body, dump):
    ins = []
    for l in body.split('\n'):
        mm = re.match(r'\s+(\S.*?)\s+//\s*([0-9A-F]+):', l)
        if mm:
            ins.append((int(mm.group(2), 16), mm.group(1)))
    addr = {a: i for i, (a, _) in enumerate(ins)}
    def target(a, t):
        mm = re.search(r's_c?branch\S*\s+(\d+)', t)
        if not mm:
            return None
        off = int(mm.group(1))
        off -= 65536 if off >= 32768 else 0
        return a + 4 + off * 4
    targets = {target(a, t) for a, t in ins} - {None}
    print(f"== {name[:140]}\n   {len(ins)} instructions, {sum(t.startswith('v_') for _, t in ins)} VALU, "
          f"{sum(t.startswith('s_') and not t.startswith(('s_waitcnt', 's_nop')) for _, t in ins)} SALU, "
          f"{sum(t.startswith('s_waitcnt') for _, t in ins)} waitcnt, {sum('readlane' in t or 'writelane' in t for _, t in ins)} SGPR-spill lane moves, "
          f"{sum(t.startswith('flat_') for _, t in ins)} flat, {sum(t.startswith('scratch_') for _, t in ins)} scratch")
    if dump:
        for i, (a, t) in enumerate(ins):
            print(i, t)
        return
    start, blocks = 0, []
    for i, (a, t) in enumerate(ins):
        if a in targets and i > start:
            blocks.append((start, i)); start = i
        if t.startswith(('s_cbranch', 's_branch', 's_endpgm')):
            blocks.append((start, i + 1)); start = i + 1
    for s, e in blocks:
        seg = ins[s:e]
        if not seg:
            continue
        v = sum(t.startswith('v_') for _, t in seg)
        sa = sum(t.startswith('s_') and not t.startswith(('s_waitcnt', 's_nop')) for _, t in seg)
        mem = [t.split()[0] for _, t in seg if t.startswith(('global_', 'buffer_', 'ds_', 'flat_', 'scratch_'))]
        tg = target(*seg[-1])
        print(f"[{s:4d}-{e:4d}) V{v:3d} S{sa:3d} mem{len(mem):2d} {','.join(sorted(set(mem)))[:70]:70s} {seg[-1][1].split()[0]} {'-> %s' % addr.get(tg, '?') if tg is not None else ''}")

if __name__ == '__main__':
    main()
