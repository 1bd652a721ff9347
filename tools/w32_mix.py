"""Instruction mix of wino32.hip's kernels from `hipcc -S` output: per variant, the interior strip copy's K phase and
epilogue.  usage: python tools/w32_mix.py wino32.s [variant substrings...]"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
names = sys.argv[2:] or ['ILi136ELi32ELi0E', 'ILi17ELi64ELi0E', 'ILi81ELi32ELi0E', 'ILi1030ELi32ELi64E']
for name in names:
    i0 = next(i for i, l in enumerate(s) if l.startswith('_ZN12_GLOBAL__N_113wino32_kernel' + name))
    end = next(i for i in range(i0, len(s)) if s[i].strip().startswith('s_endpgm'))
    body = s[i0:end]
    drains = [i for i, l in enumerate(body) if 's_nop 15' in l][::2]
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    def cnt(a, b):
        return collections.Counter(x.split()[0] for x in body[a:b] if x.startswith('\t') and not x.strip().startswith(('.', ';')))
    first2 = next(i for i in mf if i > drains[0])
    k, e = cnt(first2, drains[1]), cnt(drains[1], len(body))
    nm = sum(v for a, v in k.items() if 'mfma' in a)
    print(name, 'K text: total', sum(k.values()), 'mfma', nm, 'other', sum(k.values()) - nm, '| epilogue', sum(e.values()))
    print('  K  ', sorted(((a, v) for a, v in k.items() if 'mfma' not in a), key=lambda kv: -kv[1])[:16])
    print('  epi', sorted(e.items(), key=lambda kv: -kv[1])[:14])
