"""Per-kernel ISA statistics from a --save-temps .s file: registers, LDS, instruction counts, the waits.
    python tools/isa_stats.py <file.s> <kernel-name substring> [--loop]"""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
meta = {}
for m in re.finditer(r'\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel', s, re.S):
    if pat in m.group(1):
        b = m.group(2)
        meta[m.group(1)] = (re.findall(r'\.amdhsa_next_free_vgpr (\d+)', b), re.findall(r'\.amdhsa_accum_offset (\d+)', b),
                            re.findall(r'\.amdhsa_group_segment_fixed_size (\d+)', b), re.findall(r'\.amdhsa_private_segment_fixed_size (\d+)', b))
for m in re.finditer(r'^(\S+):\s*; @\S+\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';')]
    def cnt(p):
        return sum(1 for l in lines if re.match(p, l))
    print(name[:110])
    print('  vgpr/accum_offset/lds/scratch', meta.get(name))
    print('  instr %d  mfma %d  valu(non-mfma) %d  salu %d  ds_read %d  ds_write %d  global_load %d  glds %d  buffer %d  global_store %d  barrier %d'
          % (len(lines), cnt(r'v_mfma'), cnt(r'v_(?!mfma)'), cnt(r's_(?!waitcnt|barrier|nop)'), cnt(r'ds_read'), cnt(r'ds_write'),
             cnt(r'global_load_dword'), cnt(r'global_load_lds'), cnt(r'buffer_'), cnt(r'global_store'), cnt(r's_barrier')))
    w = [l for l in lines if l.startswith('s_waitcnt')]
    from collections import Counter
    print('  waits:', dict(Counter(w)))
