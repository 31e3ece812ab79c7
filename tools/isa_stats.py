"""Per-kernel resource and instruction-mix table of a hipcc -S --cuda-device-only listing:
    python tools/isa_stats.py file.s [name-substring]
"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+):\s*; @', s, re.M)]
for i, (pos, name) in enumerate(starts):
    if flt not in name:
        continue
    end = starts[i + 1][0] if i + 1 < len(starts) else len(s)
    seg = s[pos:end]
    k = seg.find('.amdhsa_kernel')
    if k < 0:
        continue
    body, meta = seg[:k], seg[k:]
    g = lambda key: (re.search(r'\.amdhsa_%s (\d+)' % key, meta) or [None, '?'])[1]
    cnt = lambda pat: len(re.findall(pat, body))
    print('%s\n   vgpr %s accum_off %s sgpr %s scratch %s lds %s | mfma %d (bf16 %d) valu_pk_add %d cvt_pk %d ds_read %d ds_write %d gload %d gstore %d barrier %d waitcnt %d' % (
        name[:150], g('next_free_vgpr'), g('accum_offset'), g('next_free_sgpr'), g('private_segment_fixed_size'),
        g('group_segment_fixed_size'), cnt(r'v_mfma'), cnt(r'v_mfma_f32_32x32x16_bf16'), cnt('v_pk_add_f32'),
        cnt('v_cvt_pk_bf16'), cnt(r'\bds_read'), cnt(r'\bds_write'), cnt(r'global_load'), cnt(r'global_store'),
        cnt('s_barrier'), cnt('s_waitcnt')))
