#!/usr/bin/env python3
"""Static audit of hand-counted LDS reads in the gfx950 ISA of a kernel.

The pipelined kernels request MFMA fragments with inline-asm `ds_read_b128` and wait for them with hand-counted
`s_waitcnt lgkmcnt(N)`.  hipcc does not model an asm load: it treats the destination as written at the statement and is
free to read, copy or overwrite that register before the data has landed (cdna_hip_programming.md 5.7 item 1).  A copy
of a register whose read is still in flight takes the OLD contents and shows as a rare wrong tile, depending on LDS
latency -- exactly the kind of failure that comes and goes between runs.

This tool walks the control-flow graph of one kernel of a `-S` listing, simulates the in-order LGKM queue (LDS reads,
LDS-side returns; scalar loads and s_memtime count too) along EVERY path, and reports any instruction that reads or writes
a VGPR while an LDS read into it may still be outstanding, and any counted wait (N > 0) issued while a scalar load is
pending (SMEM returns out of order: only lgkmcnt(0) covers it).

    python tools/asm_audit.py file.s|lib.so [kernel-name-substring ...]

A `-S` listing is read as it is; a shared library is taken apart first (its gfx950 code objects are extracted into a
temporary directory and disassembled with llvm-objdump), so the audit sees the instructions that ship.
Exit status 1 when a violation is found.  tests/test_host_cpu.py runs it on the hand-scheduled kernels of the product build.
"""
import os
import shutil
import subprocess
import tempfile
import re
import sys
from collections import deque

RE_LABEL = re.compile(r'^([.\w$]+):')
RE_VRANGE = re.compile(r'\bv\[(\d+):(\d+)\]')
RE_VSINGLE = re.compile(r'\bv(\d+)\b')
RE_ARANGE = re.compile(r'\ba\[(\d+):(\d+)\]')
RE_ASINGLE = re.compile(r'\ba(\d+)\b')
RE_LGKM = re.compile(r'lgkmcnt\((\d+)\)')
MAX_STATES_PER_BLOCK = 256


def regs_of(text):
    """VGPRs / AGPRs named in an operand string, as a set of ('v'|'a', index)."""
    out = set()
    for lo, hi in RE_VRANGE.findall(text):
        out.update(('v', i) for i in range(int(lo), int(hi) + 1))
    for lo, hi in RE_ARANGE.findall(text):
        out.update(('a', i) for i in range(int(lo), int(hi) + 1))
    t = RE_VRANGE.sub(' ', text)
    t = RE_ARANGE.sub(' ', t)
    out.update(('v', int(i)) for i in RE_VSINGLE.findall(t))
    out.update(('a', int(i)) for i in RE_ASINGLE.findall(t))
    return out


class Inst:
    __slots__ = ('line', 'op', 'args', 'text', 'in_asm')

    def __init__(self, line, op, args, text, in_asm):
        self.line, self.op, self.args, self.text, self.in_asm = line, op, args, text, in_asm


LLVM_BIN = os.environ.get('LP_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
RE_DIS_SYM = re.compile(r'^[0-9a-f]+ <([^>]+)>:\s*$')


def disassemble_so(path, workdir):
    """Extract the gfx950 code objects of a fat binary into workdir and disassemble them; returns the listing paths."""
    lib = os.path.join(workdir, 'lib.so')
    shutil.copy(path, lib)
    subprocess.run([os.path.join(LLVM_BIN, 'llvm-objdump'), '--offloading', lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    outs = []
    for f in sorted(os.listdir(workdir)):
        if 'amdgcn' not in f:
            continue
        dis = os.path.join(workdir, f + '.dis')
        with open(dis, 'w') as o:
            subprocess.run([os.path.join(LLVM_BIN, 'llvm-objdump'), '-d', '--symbolize-operands', '--no-show-raw-insn', os.path.join(workdir, f)],
                           check=True, stdout=o, stderr=subprocess.DEVNULL)
        outs.append(dis)
    return outs


def split_kernels_objdump(path):
    """{kernel name: (first line number, [(line number, text)])} of an llvm-objdump listing: a kernel starts at a symbol line
    `addr <name>:` whose name is not a local label `L<n>`; local labels are rewritten to the `name:` form parse() knows."""
    kernels = {}
    name, body, start = None, None, 0
    with open(path) as f:
        for n, raw in enumerate(f, 1):
            m = RE_DIS_SYM.match(raw)
            if m:
                sym = m.group(1)
                if re.fullmatch(r'L\d+', sym):
                    if name is not None:
                        body.append((n, sym + ':'))
                    continue
                if name is not None:
                    kernels[name] = (start, body)
                name, body, start = sym, [], n
                continue
            if name is not None and raw.startswith('\t'):
                body.append((n, '\t' + raw.strip().split('//')[0].rstrip()))
    if name is not None:
        kernels[name] = (start, body)
    return kernels


def split_kernels(path):
    """{kernel name: (first line number, [source lines])} of every .amdhsa kernel body in the listing."""
    with open(path) as f:
        head = f.read(4096)
    if 'file format elf64-amdgpu' in head:
        return split_kernels_objdump(path)
    kernels = {}
    name, body, start = None, None, 0
    with open(path) as f:
        for n, raw in enumerate(f, 1):
            m = RE_LABEL.match(raw)
            if m and not raw.startswith('.L') and not raw.startswith('\t'):
                lab = m.group(1)
                if name is None and '; @' in raw:
                    name, body, start = lab, [], n
                    continue
            if name is not None:
                if raw.strip().startswith('.Lfunc_end') or raw.strip().startswith('.section'):
                    kernels[name] = (start, body)
                    name = None
                    continue
                body.append((n, raw.rstrip('\n')))
    return kernels


def parse(body):
    """Instructions and label positions of a kernel body."""
    insts, labels = [], {}
    in_asm = False
    for n, raw in body:
        s = raw.strip()
        if not s:
            continue
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = RE_LABEL.match(s)
        if m and not raw.startswith('\t'):
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith(';') or s.startswith('.'):
            continue
        code = s.split(';')[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        insts.append(Inst(n, parts[0], parts[1] if len(parts) > 1 else '', code, in_asm))
    return insts, labels


def dest_regs(inst):
    """Registers the first operand names (the destination of loads / VALU)."""
    first = inst.args.split(',')[0] if inst.args else ''
    # a range operand contains a comma-free "v[a:b]" already; plain split is fine for the first operand
    return regs_of(first)


def lgkm_entries(inst):
    """How an instruction enters the LGKM queue: list of (kind, frozenset(dest regs), source line)."""
    op = inst.op
    if op.startswith('ds_read') or op.startswith('ds_load') or op.startswith('ds_bpermute') or op.startswith('ds_permute') or op.startswith('ds_swizzle'):
        return [('ds', frozenset(dest_regs(inst)), inst.line)]
    if op.startswith('ds_'):                     # LDS writes, atomics without return: counted, no destination
        return [('dsw', frozenset(), inst.line)]
    if op.startswith('s_load') or op.startswith('s_buffer_load') or op.startswith('s_memtime') or op.startswith('s_memrealtime'):
        n = 2 if op.startswith('s_mem') else 1
        return [('smem', frozenset(), inst.line)] * n
    if op.startswith('s_sendmsg') or op.startswith('s_dcache'):
        return [('smem', frozenset(), inst.line)]
    return []


def audit(insts, labels):
    """Worklist over (instruction index, queue state).  Returns a list of violation strings."""
    n = len(insts)
    # successors
    succ = [[] for _ in range(n)]
    for i, ins in enumerate(insts):
        op = ins.op
        if op == 's_endpgm':
            continue
        if op == 's_branch':
            t = labels.get(ins.args.strip())
            if t is not None and t < n:
                succ[i].append(t)
            continue
        if op.startswith('s_cbranch'):
            t = labels.get(ins.args.strip())
            if t is not None and t < n:
                succ[i].append(t)
        if op == 's_setpc_b64' or op == 's_swappc_b64':
            continue
        if i + 1 < n:
            succ[i].append(i + 1)
    # block leaders = targets and fall-throughs after branches; we simply memoise states at every label target and loop on instructions
    leaders = set(labels.values()) | {0}
    seen = {}           # leader index -> set of states
    violations = {}
    work = deque([(0, ())])
    while work:
        i, q = work.popleft()
        q = list(q)
        while True:
            if i in leaders:
                st = tuple(q)
                s = seen.setdefault(i, set())
                if st in s:
                    break
                if len(s) >= MAX_STATES_PER_BLOCK:
                    violations.setdefault(('states', i), 'line %d: more than %d distinct LGKM queue states at this label (audit gave up on it)' % (insts[i].line, MAX_STATES_PER_BLOCK))
                    break
                s.add(st)
            ins = insts[i]
            pending = set()
            for kind, regs, _ln in q:
                pending |= regs
            m = RE_LGKM.search(ins.text) if ins.op == 's_waitcnt' else None
            if ins.op == 's_waitcnt':
                if m:
                    keep = int(m.group(1))
                    if keep > 0 and any(k == 'smem' for k, _, _ln in q):
                        violations.setdefault(('smem', ins.line), 'line %d: `%s` while a scalar load may be pending (SMEM returns out of order)' % (ins.line, ins.text))
                    if len(q) > keep:
                        q = q[len(q) - keep:] if keep else []
            else:
                touched = regs_of(ins.args)
                ent = lgkm_entries(ins)
                hit = touched & pending
                if hit:
                    who = ', '.join('%s%d' % r for r in sorted(hit))
                    src = sorted({ln for _k, regs, ln in q if regs & hit})
                    violations.setdefault(('reg', ins.line), 'line %d: `%s` touches %s while an LDS read into it (issued at line %s) may be in flight' % (
                        ins.line, ins.text, who, ', '.join(map(str, src))))
                q.extend(ent)
                if len(q) > 64:
                    q = q[-64:]
            nxt = succ[i]
            if not nxt:
                break
            for t in nxt[1:] if len(nxt) > 1 and nxt[0] != i + 1 else []:
                pass
            if len(nxt) == 1:
                i = nxt[0]
                continue
            # conditional branch: target first in list, fall-through second
            work.append((nxt[0], tuple(q)))
            i = nxt[1]
    return [violations[k] for k in sorted(violations, key=lambda k: (str(k[0]), k[1]))]


def audit_file(path, patterns=()):
    """{kernel: (instructions, [violations])} of a -S listing, an objdump listing or a shared library."""
    with open(path, 'rb') as f:
        is_elf = f.read(4) == b'\x7fELF'
    if is_elf:
        res = {}
        tmp = tempfile.mkdtemp(prefix='lp_audit_')
        try:
            for dis in disassemble_so(path, tmp):
                res.update(audit_file(dis, patterns))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return res
    res = {}
    for name, (start, body) in split_kernels(path).items():
        if patterns and not any(p in name for p in patterns):
            continue
        insts, labels = parse(body)
        res[name] = (len(insts), audit(insts, labels))
    return res


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    res = audit_file(argv[1], argv[2:])
    bad = 0
    for name, (ninst, v) in res.items():
        print('%s: %d instructions, %d violations' % (name, ninst, len(v)))
        for s in v[:40]:
            print('   ', s)
        bad += len(v)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
