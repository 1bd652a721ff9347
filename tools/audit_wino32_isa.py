#!/usr/bin/env python3
"""Audit of the hand-placed MFMA statements of wino32.hip in the compiler's ISA (cdna_hip_programming.md 5.7 items 2, 4):
1. inside the K loop of every wino32_kernel instantiation no compiler-generated instruction may touch an accumulator register
   (the compiler does not know the MFMA latencies of an asm statement) and no scratch access may appear;
2. every MFMA has at least two wait states between it and the last vector instruction that wrote one of its source operands
   (A, B or the accumulator).  That is the gfx950 hazard behind the run-to-run wrong accumulators round 3 saw without the
   `s_nop 1` at the head of each statement: hipcc's own hazard recognizer pads exactly this for a builtin MFMA (a v_mul
   writing the A operand in front of __builtin_amdgcn_mfma_f32_16x16x4f32 comes out as v_mul, s_waitcnt, s_nop 0, v_mfma -
   two wait states - tools/mfma_operand_hazard.hip reproduces it), and pads nothing for an asm statement it cannot see into.
usage: hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only wino32.hip -o wino32.s; python tools/audit_wino32_isa.py wino32.s
(tests/test_host_cpu.py::test_wino32_isa_audit runs this on every CPU test pass)"""
import re
import sys


def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def wait_states(l):
    """Wait states an instruction contributes in front of a later one: s_nop N = N + 1, anything else that issues = 1."""
    m = re.match(r's_nop\s+(\d+)', l)
    return int(m.group(1)) + 1 if m else 1


def is_instr(l):
    return bool(l) and not l.startswith((';', '.')) and not l.endswith(':')


def operand_hazards(body):
    """MFMAs whose source operands were written by a vector (non-MFMA) instruction fewer than two wait states earlier."""
    instrs = [l.strip() for l in body if is_instr(l.strip())]
    bad = []
    for i, l in enumerate(instrs):
        if not l.startswith('v_mfma'):
            continue
        ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
        srcs = set()
        for t in ops[1:]:
            srcs |= regs_of(t)
        ws, j = 0, i - 1
        while j >= 0 and ws < 2:
            p = instrs[j]
            if p.startswith('v_') and not p.startswith('v_mfma'):
                dst = p.split(None, 1)[1].split(',')[0].strip() if ' ' in p else ''
                if regs_of(dst) & srcs:
                    bad.append((p, l))
                    break
            ws += wait_states(p)
            j -= 1
    return bad


def main(path):
    lines = open(path).read().split('\n')
    bad = 0
    starts = [i for i, l in enumerate(lines) if re.match(r'_ZN\S*wino32_kernel\S*:', l)]
    for st in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].strip().startswith('s_endpgm'))
        body = lines[st:end]
        in_asm = False
        mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
        # split into strip bodies: a gap of > 600 lines between MFMAs separates the copies / the epilogue
        groups, cur = [], [mf[0]]
        for a, b in zip(mf, mf[1:]):
            if b - a > 400:
                groups.append(cur)
                cur = []
            cur.append(b)
        groups.append(cur)
        nscr = 0
        for g in groups:
            live = set()  # accumulator registers written by an MFMA so far in this K loop (before that they are free)
            viol = []
            in_asm = False
            for i in range(max(0, g[0] - 4), g[-1] + 1):
                l = body[i].strip()
                if l.startswith(';;#ASMSTART'):
                    in_asm = True
                elif l.startswith(';;#ASMEND'):
                    in_asm = False
                elif 'v_mfma' in l:
                    assert in_asm, 'an MFMA outside an asm statement: ' + l
                    live |= regs_of(l.split()[1].rstrip(','))
                elif not in_asm and l and not l.startswith(';') and not l.startswith('.'):
                    if 'scratch_' in l:
                        nscr += 1
                    toks = re.findall(r'v\[\d+:\d+\]|v\d+', l)
                    if any(regs_of(t) & live for t in toks):
                        viol.append((i, l))
            if viol:
                bad += len(viol)
                print(lines[st][:70], 'K loop', g[0], '-', g[-1], ':', len(viol), 'compiler instructions touch accumulators, e.g.',
                      viol[0])
        hz = operand_hazards(body)
        if hz:
            print(lines[st][:70], ':', len(hz), 'MFMAs read an operand less than two wait states behind its writer, e.g.', hz[0])
        print(lines[st][:80], ':', len(mf), 'MFMAs in', len(groups), 'K loops, scratch accesses inside them:', nscr,
              ', operand hazards:', len(hz))
        bad += nscr + len(hz)
    print('AUDIT', 'FAILED' if bad else 'ok')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
