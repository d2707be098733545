#!/usr/bin/env python3
"""What HIP-graph capture on this ROCm tolerates around events (each case in a child process: the bad ones end in a segmentation fault):
an event recorded on the capturing stream and destroyed before the capture ends / kept alive; a fork to a second stream and back with
the events destroyed at once (torch's Stream.wait_stream) / kept alive; a multi-stream autograd backward inside a capture."""
import subprocess
import sys

CASES = {
    'record_destroy': '''
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    y = x * 2
    e = torch.cuda.Event(); e.record(); del e
    y = y + 1
''',
    'record_keep': '''
g = torch.cuda.CUDAGraph()
keep = []
with torch.cuda.graph(g, stream=s):
    y = x * 2
    e = torch.cuda.Event(); e.record(); keep.append(e)
    y = y + 1
''',
    'fork_join_wait_stream': '''
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    y = x * 2
    side.wait_stream(s)
    with torch.cuda.stream(side):
        z = x * 3
    s.wait_stream(side)
    y = y + z
''',
    'fork_join_keep_events': '''
g = torch.cuda.CUDAGraph()
keep = []
with torch.cuda.graph(g, stream=s):
    y = x * 2
    e = torch.cuda.Event(); e.record(s); side.wait_event(e); keep.append(e)
    with torch.cuda.stream(side):
        z = x * 3
        e2 = torch.cuda.Event(); e2.record(side); keep.append(e2)
    s.wait_event(e2)
    y = y + z
''',
    'autograd_two_streams': '''
w = torch.randn(256, 256, device='cuda', requires_grad=True)
w2 = torch.randn(256, 256, device='cuda', requires_grad=True)
def step():
    a = x @ w
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        b = (x @ w2).relu()
    torch.cuda.current_stream().wait_stream(side)
    (a.sum() + b.sum()).backward()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
w.grad = None; w2.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
g.replay(); torch.cuda.synchronize()
print(float(w.grad.sum()), float(w2.grad.sum()))
''',
}
PRE = '''
import torch
x = torch.randn(256, 256, device='cuda')
s = torch.cuda.Stream(); side = torch.cuda.Stream()
torch.cuda.synchronize()
'''
POST = '''
g.replay(); torch.cuda.synchronize(); print('ok')
'''
for name, body in CASES.items():
    r = subprocess.run([sys.executable, '-c', PRE + body + POST], capture_output=True, text=True)
    print(f'{name:28s} rc={r.returncode} {r.stdout.strip()[-80:]} {r.stderr.strip()[-200:] if r.returncode else ""}', flush=True)
