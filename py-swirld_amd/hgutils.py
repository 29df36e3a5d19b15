"""Graph helpers of the gossip side (counterparts of /root/reference/utils.py:8-55)."""
from collections import deque

from . import crypto


def toposort(nodes, parents):
    """Yield `nodes` parents-first (utils.py:8-21); ValueError on a cycle."""
    WHITE, GREY, BLACK = 0, 1, 2
    state = {}
    out = []
    for root in nodes:
        if state.get(root, WHITE) != WHITE:
            continue
        stack = [(root, iter(parents(root)))]
        state[root] = GREY
        while stack:
            u, it = stack[-1]
            advanced = False
            for v in it:
                if v not in nodes:
                    continue
                s = state.get(v, WHITE)
                if s == GREY:
                    raise ValueError("not a DAG")
                if s == WHITE:
                    state[v] = GREY
                    stack.append((v, iter(parents(v))))
                    advanced = True
                    break
            if not advanced:
                state[u] = BLACK
                out.append(u)
                stack.pop()
    return out


def bfs(sources, succ):
    """Breadth-first traversal (utils.py:24-34)."""
    sources = tuple(sources)
    seen = set(sources)
    q = deque(sources)
    while q:
        u = q.popleft()
        yield u
        for v in succ(u):
            if v not in seen:
                seen.add(v)
                q.append(v)


def randrange(n):
    """Uniform integer in [0, n) by rejection sampling on random bytes (utils.py:49-55)."""
    nbytes = (n.bit_length() + 7) // 8
    shift = 8 * nbytes - n.bit_length()
    while True:
        r = int.from_bytes(crypto.randombytes(nbytes), "big") >> shift
        if r < n:
            return r
