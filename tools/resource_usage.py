"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: one row per kernel
(VGPRs, AGPRs, scratch bytes/lane, occupancy, LDS).  Usage:
    python -c "from hybridq_amd import build as b; log=[]; b.build(force=True, extra_flags=['-Rpass-analysis=kernel-resource-usage'], lib='/tmp/x.so', objdir='/tmp/xobj', log=log); open('ru.txt','w').write('\\n'.join(log))"
    python tools/resource_usage.py ru.txt
"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r'remark:\s+(.*?): (\S+) \[-Rpass-analysis', line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == 'Function Name':
            cur = {'name': val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


def main():
    rows = parse(open(sys.argv[1]).read())
    names = demangle([r['name'] for r in rows])
    print('kernel,vgprs,agprs,scratch_bytes_per_lane,occupancy_waves_per_simd,sgprs,lds_bytes')
    for r, n in zip(rows, names):
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'\(.*$', '', n).replace('hq::', '')
        print(','.join(['"' + n + '"', r.get('VGPRs', ''), r.get('AGPRs', ''), r.get('ScratchSize [bytes/lane]', ''),
                        r.get('Occupancy [waves/SIMD]', ''), r.get('TotalSGPRs', ''), r.get('LDS Size [bytes/block]', '')]))


if __name__ == '__main__':
    main()
