import re,sys
f=sys.argv[1]; name=sys.argv[2]
lines=open(f).read().split('\n')
start=[i for i,l in enumerate(lines) if l.startswith(name+':')][0]
end=[i for i,l in enumerate(lines) if i>start and 's_endpgm' in l][0]
seg=lines[start:end]
keys=['mfma','valu','salu','ds_r','ds_w','buf','glob','scr_ld','scr_st','acc_mov','wait','nop']
cnt={k:0 for k in keys}
def flush(tag):
    global cnt
    print(tag,{k:v for k,v in cnt.items() if v})
    cnt={k:0 for k in keys}
for i,l in enumerate(seg):
    t=l.strip()
    if t.startswith('s_barrier'): flush(f'-- barrier at {i}')
    elif t.startswith('v_mfma'): cnt['mfma']+=1
    elif t.startswith('scratch_load'): cnt['scr_ld']+=1
    elif t.startswith('scratch_store'): cnt['scr_st']+=1
    elif t.startswith('v_accvgpr'): cnt['acc_mov']+=1
    elif t.startswith('ds_read') or t.startswith('ds_load'): cnt['ds_r']+=1
    elif t.startswith('ds_write') or t.startswith('ds_store'): cnt['ds_w']+=1
    elif t.startswith('buffer_'): cnt['buf']+=1
    elif t.startswith('global_'): cnt['glob']+=1
    elif t.startswith('s_waitcnt'): cnt['wait']+=1
    elif t.startswith('s_nop'): cnt['nop']+=1
    elif t.startswith('v_'): cnt['valu']+=1
    elif t.startswith('s_'): cnt['salu']+=1
flush('-- end')
