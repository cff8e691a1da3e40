"""Copy the evidence of scratch/run_final.sh <tag> from gpurun_out/ into profiles/ (tracked) under the round's names and install the
PMC aggregate as profiles/traffic.json.  usage: python scratch/collect_profiles.py <tag>"""
import os, shutil, sys
tag = sys.argv[1]
g, p = 'gpurun_out', 'profiles'
pairs = [(f'{g}/kernel_stats_{tag}_serial.csv', f'{p}/{tag}_kernel_stats_serial.csv'), (f'{g}/kernel_stats_{tag}.csv', f'{p}/{tag}_kernel_stats.csv'),
         (f'{g}/kernel_stats_train_{tag}.csv', f'{p}/{tag}_kernel_stats_train.csv'), (f'{g}/kernel_stats_train_{tag}_dist.csv', f'{p}/{tag}_kernel_stats_train_dist.csv'),
         (f'{g}/{tag}/bench_default.json', f'{p}/{tag}_bench_default.json'), (f'{g}/{tag}/bench_serial.json', f'{p}/{tag}_bench_serial.json'),
         (f'{g}/{tag}/bench_squeezedetplus.json', f'{p}/{tag}_bench_squeezedetplus.json'), (f'{g}/traffic_{tag}.json', f'{p}/{tag}_traffic_pmc.json'),
         (f'{g}/traffic_{tag}.json', f'{p}/traffic.json'), (f'{g}/{tag}/fuzz_conv.log', f'{p}/{tag}_fuzz_conv.log'), (f'{g}/{tag}/smoke.log', f'{p}/{tag}_smoke.log')]
for a, b in pairs:
    if os.path.exists(a):
        shutil.copyfile(a, b); print('copied', b)
    else:
        print('MISSING', a)
log = f'{g}/{tag}/pytest_gpu.log'
if os.path.exists(log):
    open(f'{p}/{tag}_pytest_gpu.log', 'w').write(''.join(open(log).readlines()[-3:]))
