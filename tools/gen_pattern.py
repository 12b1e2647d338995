"""Regenerates orb_pattern.inc (the published 256-pair rBRIEF table) from a text
source that lists the 1024 integers; comments are stripped.  Usage:
    python tools/gen_pattern.py <file-with-1024-ints> 
Writes oracle/orb_pattern.inc and ssvio_amd/csrc/orb_pattern.inc."""
import re, sys, pathlib
root = pathlib.Path(__file__).resolve().parent.parent
src = open(sys.argv[1]).read()
s = src.index('bit_pattern_31_'); s = src.index('{', s); e = src.index('};', s)
body = re.sub(r'/\*.*?\*/', '', src[s + 1:e], flags=re.S)
nums = [int(x) for x in re.findall(r'-?\d+', body)]
assert len(nums) == 1024 and max(map(abs, nums)) <= 13
old = (root / 'oracle/orb_pattern.inc').read_text()
hdr = old[:old.index('*/') + 3]
txt = hdr + ',\n'.join(','.join('%d' % v for v in nums[i:i + 32]) for i in range(0, 1024, 32)) + '\n'
for p in ('oracle/orb_pattern.inc', 'ssvio_amd/csrc/orb_pattern.inc'):
    (root / p).write_text(txt)
