#!/usr/bin/env python3
"""How does a replayed HIP graph run two independent branches?  Branch A = one long kernel (a ~1 ms spin), branch B = 30 short dependent
kernels; captured in different orders / on different streams; replay time says whether they ran side by side (~1 ms) or one after the
other (~1.3+ ms).  The same fork / join enqueued eagerly for comparison."""
import time

import torch

x = torch.zeros(1 << 16, device='cuda')
SPIN = 2_000_000   # torch.cuda._sleep cycles, ~1 ms


def branch_a():
    torch.cuda._sleep(SPIN)


def branch_b():
    y = x
    for _ in range(30):
        y = y + 1.0
    return y


def run(order, a_on, b_on, origin, side, side2):
    streams = {'origin': origin, 'side': side, 'side2': side2}

    def on(which, fn):
        s = streams[which]
        if s is origin:
            return fn()
        s.wait_stream(origin)
        with torch.cuda.stream(s):
            r = fn()
        return r

    def body():
        for which in order:
            if which == 'a':
                on(a_on, branch_a)
            else:
                on(b_on, branch_b)
        for w in {a_on, b_on}:
            if streams[w] is not origin:
                origin.wait_stream(streams[w])
    return body


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


origin, side, side2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
torch.cuda.synchronize()
with torch.cuda.stream(origin):
    a_ms = timeit(branch_a)
    b_ms = timeit(branch_b)
print(f'alone: A {a_ms:.3f} ms, B {b_ms:.3f} ms')
for order, a_on, b_on in ((('b', 'a'), 'origin', 'side'), (('a', 'b'), 'origin', 'side'), (('b', 'a'), 'side2', 'side'), (('a', 'b'), 'side', 'origin'),
                          (('b', 'a'), 'side', 'origin'), (('a', 'b'), 'origin', 'side2')):
    body = run(order, a_on, b_on, origin, side, side2)
    with torch.cuda.stream(origin):
        eager = timeit(body)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=origin):
        body()
    replay = timeit(g.replay)
    print(f'capture order {order}, A on {a_on:6s}, B on {b_on:6s}: eager {eager:.3f} ms, replay {replay:.3f} ms')
