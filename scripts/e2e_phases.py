"""Where save_embedded_obs.run's wall time goes on the bench's synthetic scene (bench.py::save_obs_e2e_bench): wraps the phases with timers.
Usage: python scripts/e2e_phases.py [n_samples]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pvr_habitat_amd import save_embedded_obs as S, embeddings as E, scene_pickle as SP

T = collections.OrderedDict()


def timed(mod, name, key=None):
    fn = getattr(mod, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            T[key or name] = T.get(key or name, 0.0) + time.perf_counter() - t0
    setattr(mod, name, w)


timed(SP, 'scene_index'); timed(S, 'stitch_shards'); timed(S, 'stream_embed'); timed(S, 'load_complete_shard'); timed(S, 'EmbeddingNet', 'EmbeddingNet()')
timed(S.ShardWriter, 'append', 'writer.append'); timed(S.ShardWriter, 'finish', 'writer.finish')
timed(SP, 'scene_rows', 'scene_rows (reader thread)')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
res = bench.save_obs_e2e_bench(256, 'bf16', n_samples=n)
print({k: res[k] for k in ('value', 'wall_s', 'scene_MB', 'out_MB')})
for k, v in T.items():
    print('  %-28s %.2f s (both runs: warm-up + timed)' % (k, v))
if len(sys.argv) > 2:
    for si in (0.005, 0.0005, 0.0001):
        sys.setswitchinterval(si)
        T.clear()
        res = bench.save_obs_e2e_bench(256, 'bf16', n_samples=n)
        print('switch interval %.4f: %s' % (si, {k: res[k] for k in ('value', 'wall_s')}), {k: round(v, 2) for k, v in T.items()})
