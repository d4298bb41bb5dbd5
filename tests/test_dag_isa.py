"""The emitted ISA of the control steps of the single-launch factorisation (csrc/dag.hpp).

Round 2 lost a day to a compiler transformation: with wait / publish / ticket fetch inlined into the worker loop, hipcc merged
the spin loops of the two task kinds and sank a counter store out of the loop that must publish it -- the kernel hung on the
GPU.  The fix (control steps as __noinline__ calls) holds only as long as the compiler keeps emitting what the protocol needs,
so this test compiles the library to assembly (`make -C cosmogp_amd/csrc asm`, no GPU involved) and checks, function by
function, the invariants the visibility argument of dag.hpp rests on:

  dag_publish    s_waitcnt vmcnt(0)  ->  s_barrier  ->  ONE global store of the counter with sc1 (agent scope), nothing else
                 stored; it is a real function (not inlined away)
  dag_wait /     the poll is a global_load_dword ... sc1 INSIDE a loop (a backward branch follows it), the abort word is
  dag_wait_opt   polled the same way, the result travels through LDS and an s_barrier
  dag_fetch      the ticket is ONE global_atomic_add ... sc0 (returning) and an s_barrier follows
  k_chol_dag     calls (s_swappc_b64) them: at least one call site per control step
  hand-off data  every 16-byte load / store of dag_upd_tile and of the solves is a buffer_* instruction with sc1
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cosmogp_amd', 'csrc')


@pytest.fixture(scope='module')
def asm():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.check_call(['make', '-C', CSRC, 'asm', 'HIPCC=%s' % hipcc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(os.path.join(CSRC, 'api.s')) as f:
        text = f.read()
    funcs = {}
    for m in re.finditer(r'^(_Z[\w]+):.*?\n(.*?)^\s*\.size\s+\1', text, re.S | re.M):
        body = [l.split(';')[0].strip() for l in m.group(2).splitlines()]
        funcs[m.group(1)] = [l for l in body if l and not l.startswith('.') or l.startswith('.LBB')]
    return funcs


def _one(funcs, prefix):
    names = [n for n in funcs if re.match(r'_Z\d+%s(?![a-z_])' % prefix, n)]
    assert len(names) == 1, (prefix, names)
    return funcs[names[0]]


def _index(body, pattern, start=0):
    for i in range(start, len(body)):
        if re.search(pattern, body[i]):
            return i
    return -1


@pytest.mark.timeout(900)
def test_publish_orders_drain_barrier_store(asm):
    body = _one(asm, 'dag_publish')
    w = _index(body, r'^s_waitcnt .*vmcnt\(0\)')
    b = _index(body, r'^s_barrier', w)
    st = _index(body, r'^global_store_dword\b.* sc1', b)
    assert 0 <= w < b < st, body
    stores = [l for l in body if re.match(r'(global|flat|buffer|scratch)_store|global_atomic', l)]
    assert len(stores) == 1, stores                      # the counter, nothing else
    assert not any(re.match(r'^s_barrier', l) for l in body[st:])


@pytest.mark.timeout(900)
@pytest.mark.parametrize('fn', ['dag_wait', 'dag_wait_opt'])
def test_wait_polls_with_sc1_inside_a_loop(asm, fn):
    body = _one(asm, fn)
    polls = [i for i, l in enumerate(body) if re.match(r'^global_load_dword\b.* sc1', l)]
    assert len(polls) >= 2, body                          # the counters and the abort word
    labels = {l[:-1]: i for i, l in enumerate(body) if l.startswith('.LBB') and l.endswith(':')}
    first = polls[0]
    # a backward branch to a label at or before the first poll closes the spin loop
    back = [i for i, l in enumerate(body) if i > first and re.match(r'^s_cbranch_\w+ (\.LBB\w+)', l)
            and labels.get(l.split()[-1], 10 ** 9) <= first]
    assert back, 'the counter poll is not inside a loop'
    assert all(not re.match(r'^(global|flat)_load_dword\b(?!.* sc1)', l) for l in body), 'a poll without sc1'
    assert _index(body, r'^s_barrier', back[0]) > 0 and _index(body, r'^ds_write_b32') > 0
    assert not any(re.match(r'^buffer_(wbl2|inv)', l) for l in body)      # no fences: the protocol is sc1 end to end


@pytest.mark.timeout(900)
def test_fetch_is_one_returning_atomic(asm):
    body = _one(asm, 'dag_fetch')
    at = [i for i, l in enumerate(body) if re.match(r'^global_atomic_add\b.* sc0', l)]
    assert len(at) == 1, body
    assert _index(body, r'^s_barrier', at[0]) > 0


@pytest.mark.timeout(900)
def test_kernel_calls_the_control_steps_and_hands_data_off_through_sc1(asm):
    kern = _one(asm, 'k_chol_dag')
    assert sum(1 for l in kern if l.startswith('s_swappc_b64')) >= 5      # fetch, two waits, two publishes (+ the task bodies)
    wide = [l for l in kern if re.match(r'^(buffer|global|flat)_(load|store)_dwordx4', l)]
    other = [l for l in wide if not (l.startswith('buffer_') and ' sc1' in l)]
    # (the only 16-byte accesses that are not sc1 buffer instructions: the two stores of the optional profiling words)
    assert wide and all(l.startswith('global_store_dwordx4') for l in other) and len(other) <= 2, other[:5]
    for fn in ('dag_trsm_task', 'dag_leaf_step'):
        body = _one(asm, fn)
        wide = [l for l in body if re.match(r'^(buffer|global|flat)_(load|store)_dwordx4', l)]
        assert wide and all(l.startswith('buffer_') and ' sc1' in l for l in wide), fn
