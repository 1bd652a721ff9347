#!/usr/bin/env python3
"""Audit of the hand-placed MFMA statements of wino32.hip in the compiler's ISA (cdna_hip_programming.md 5.7 item 4): inside
the K loop of every wino32_kernel instantiation no compiler-generated instruction may touch an accumulator register (the
compiler does not know the MFMA latencies of an asm statement) and no scratch access may appear.
usage: hipcc ... -save-temps -c wino32.hip; python tools/audit_wino32_isa.py wino32-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys


def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


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
                    # (an A / B operand written by the vector instruction just before needs no wait states: hipcc's own
                    # hazard model - GCNHazardRecognizer::checkMAIHazards90A - has none for it either)
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
        print(lines[st][:80], ':', len(mf), 'MFMAs in', len(groups), 'K loops, scratch accesses inside them:', nscr)
        bad += nscr
    print('AUDIT', 'FAILED' if bad else 'ok')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
