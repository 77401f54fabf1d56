"""How do HIP streams map onto hardware queues, and what does it do to stream_embed?  Pinned source, 32 batches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvr_habitat_amd import synth
import pvr_habitat_amd.embeddings as E
class Net: pass
net = Net(); net.embedding = E.HipResNet50(synth.resnet50_state_dict(1, 'conv5'), 'conv5', compute_dtype='bf16', max_batch=256); net.out_size = 2048
fr = torch.from_numpy(synth.frames(1, 2048, 256, 256)).repeat(4, 1, 1, 1).pin_memory()
out = torch.empty((fr.shape[0], 2048), dtype=torch.float32, pin_memory=True)
def rate(label):
    E.stream_embed(net, fr[:1024], 256, out=out[:1024])
    t0 = time.perf_counter(); E.stream_embed(net, fr, 256, out=out); el = time.perf_counter() - t0
    print('%-70s %.0f frames/s' % (label, fr.shape[0] / el), flush=True)
for order in ('hdab', 'abhd', 'ahbd', 'xhdab', 'xabhd', 'xxhdab', 'haxdb', 'hadb', 'xhadb'):
    E._STREAM_CACHE.clear()
    os.environ['PVR_STREAM_ORDER'] = order
    rate('creation order %s' % order)
