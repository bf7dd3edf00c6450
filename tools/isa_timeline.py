"""Condensed instruction-class listing of one kernel in a hipcc -S output: runs of MFMA / VALU / LDS / global ops with every
s_waitcnt kept verbatim -- shows where a wave waits for memory.    python tools/isa_timeline.py file.s [kernel-index]"""
import sys


def cls(l):
    l = l.strip()
    if not l or l[0] in ';.':
        return None
    op = l.split()[0]
    if op.endswith(':'):
        return '\n' + op
    for pre, name in (('v_mfma', 'mfma'), ('global_load', 'gload'), ('global_store', 'gstore'), ('scratch_load', 'SCRL'),
                      ('scratch_store', 'SCRS'), ('ds_read', 'dsr'), ('ds_load', 'dsr'), ('ds_write', 'dsw'), ('ds_store', 'dsw'),
                      ('s_load', 'sload'), ('s_barrier', 'BARRIER'), ('buffer_load', 'bload')):
        if op.startswith(pre):
            return name
    if op.startswith('s_waitcnt') or op.startswith('s_cbranch') or op.startswith('s_branch'):
        return '[' + ' '.join(l.split()[:3]).replace('s_waitcnt ', 'W ').replace('s_cbranch_', 'br_') + ']'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_endpgm'):
        return 'END'
    return None


def main():
    lines = open(sys.argv[1]).read().split('\n')
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kern, out, prev, cnt = -1, [], None, 0
    for l in lines:
        if l.startswith('_Z') and ':' in l:
            kern += 1
        if kern != want:
            continue
        c = cls(l)
        if c is None:
            continue
        if c == prev:
            cnt += 1
        else:
            if prev:
                out.append(prev + ('x%d' % cnt if cnt > 1 else ''))
            prev, cnt = c, 1
        if c == 'END':
            break
    out.append(prev)
    print(' '.join(out))


if __name__ == '__main__':
    main()
