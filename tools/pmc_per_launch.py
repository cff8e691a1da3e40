"""Per-LAUNCH hardware counters of one bench.py step ("per-launch FETCH/WRITE of the training conv_wino launches,
not the per-kernel average"; "TCP/TCC counters for the 1x1 family").  Runs rocprofv3 --pmc once per counter group over
`bench.py --mode <mode> --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-pipeline`, then averages every counter over the eager
steps per (kernel, k-th launch of that kernel inside a step) and tags the rows with the host-side launch plan (plan.py).
usage (GPU box, repo root):  python tools/pmc_per_launch.py <tag> <infer|train> "<group1 counters>" ["<group2 counters>" ...]
 -> gpurun_out/pmc_launch_<tag>_<mode>.json / .txt"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scratch'))


def short_name_fn():
    src = open(os.path.join(ROOT, 'scratch', 'traffic_aggregate.py')).read()
    ns = {}
    exec(src[src.index('import csv'):src.index('def aggregate(')], ns)         # (only the name-mapping helper)
    return ns['short_name']


def main():
    tag, mode, groups = sys.argv[1], sys.argv[2], sys.argv[3:]
    short_name = short_name_fn()
    env = dict(os.environ, TMPDIR='/tmp')
    data = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> [values in dispatch order]
    steps = None
    for gi, grp in enumerate(groups):
        out = os.path.join(ROOT, 'gpurun_out', f'pmcl_{tag}_{mode}_{gi}')
        os.makedirs(out, exist_ok=True)
        cmd = ['rocprofv3', '--pmc'] + grp.split() + ['--output-format', 'csv', '-d', out, '-o', 'pmc', '--', 'python3', os.path.join(ROOT, 'bench.py'),
               '--mode', mode, '--steps', '3', '--warmup', '2', '--no-cpu-baseline', '--no-graph', '--no-pipeline']
        with open(os.path.join(out, 'out.json'), 'w') as fo, open(os.path.join(out, 'stderr.log'), 'w') as fe:
            rc = subprocess.run(cmd, cwd='/tmp', env=env, stdout=fo, stderr=fe, timeout=400).returncode
        if rc != 0:
            print(f'group {gi} ({grp}): rocprofv3 rc {rc}', open(os.path.join(out, 'stderr.log')).read()[-400:])
            continue
        line = json.loads([l for l in open(os.path.join(out, 'out.json')) if l.startswith('{')][-1])
        steps = line['eager_steps_launched'] if mode == 'infer' else line.get('eager_steps_launched', line.get('train', {}).get('eager_steps_launched'))
        f = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)[0]
        rows = list(csv.DictReader(open(f)))
        per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch id -> counter -> value
        for r in rows:
            per[short_name(r['Kernel_Name'])][int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        for k, disp in per.items():
            for d in sorted(disp):
                for c, v in disp[d].items():
                    data[k][c].append(v)
    from squeezedet_pytorch_amd import plan
    p = plan.inference_launch_plan() if mode == 'infer' else plan.training_launch_plan()
    tags = collections.defaultdict(list)
    for name, t in p:
        tags[name].append(t)
    result = {'_meta': {'mode': mode, 'eager_steps': steps, 'groups': groups,
                        'how': 'rocprofv3 --pmc, one pass per group; value = mean over the eager steps of the k-th launch of the kernel inside a step'}}
    lines = []
    for k, cs in data.items():
        n = max(len(v) for v in cs.values())
        if not steps or n % steps:
            continue
        per_step = n // steps
        for j in range(per_step):
            row = {c: sum(v[j::per_step]) / len(v[j::per_step]) for c, v in cs.items() if len(v) == n}
            t = tags.get(k, [])
            row_tag = (t[j] if len(t) == per_step else '?') + f' #{j}'
            result[f'{k} | {row_tag}'] = row
    keys = sorted({c for k, v in result.items() if k != '_meta' for c in v})
    lines.append('launch'.ljust(64) + ''.join(c[-22:].rjust(24) for c in keys))
    for k, v in result.items():
        if k == '_meta':
            continue
        lines.append(k[:64].ljust(64) + ''.join((f'{v[c]:.4g}' if c in v else '-').rjust(24) for c in keys))
    base = os.path.join(ROOT, 'gpurun_out', f'pmc_launch_{tag}_{mode}')
    json.dump(result, open(base + '.json', 'w'), indent=1)
    open(base + '.txt', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:60]))


if __name__ == '__main__':
    main()
