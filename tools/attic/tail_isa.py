"""Development aid: the memory / wait instructions of k_chain_persist<1,1024>'s tail (after its last MFMA) from build/dis."""
import re, sys
s = open('/root/repo/build/dis/kernels_chain.s').read()
i = s.index('_Z15k_chain_persistILi1ELi1024ELb0ELb0EEvPK4ViewiiiijP9ChainSyncPjii6HoWork6XcWorkij:')
j = s.index('.end_amdhsa_kernel', i)
body = s[i:j].split('\n')
for k in ('next_free_vgpr', 'next_free_sgpr', 'private_segment_fixed_size'):
    print(k, re.search(r'\.amdhsa_' + k + r' (\S+)', s[i:j]).group(1))
last = max(k for k, l in enumerate(body) if 'v_mfma' in l)
print('lines', len(body), 'last mfma', last)
n = 0
lim = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for k in range(last, len(body)):
    l = body[k]
    if re.search(r's_load|s_waitcnt|s_barrier|buffer_load|global_load|flat_load|v_readlane|v_writelane|s_cbranch|scratch', l):
        print(k, l.strip()[:100]); n += 1
    if n > lim: break
