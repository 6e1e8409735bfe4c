"""Paged-KV block bookkeeping: the producer of the `kv_cache_block_pointers` tensor GPTAttention consumes.
Mirror of tensorrt_llm/runtime/kv_cache_manager.py:20-387 (Block, GenerationSequence, BlocksManager, KVCacheManager):
same class and method names, same pool layout (a pool holds all K blocks, then all V blocks: K pointer =
base + idx * block_bytes, V pointer = K pointer + blocks * block_bytes, :186-193, :100-110), same allocation order
(free blocks are handed out first-in first-out) and the same pointer tensor `[pools, batch, beam, 2, max_blocks]`.
Host-only Python, like the reference's."""
from collections import deque

import numpy as np
import torch

_SIZEOF = {torch.float32: 4, torch.float16: 2, torch.bfloat16: 2, torch.int8: 1, torch.uint8: 1}
if hasattr(torch, "float8_e4m3fn"):
    _SIZEOF[torch.float8_e4m3fn] = 1


class Block:
    """One cache block: its index in every pool and a reference count (shared across beams while > 1)."""

    __slots__ = ("idx", "k_ptrs", "v_ptrs", "ref_count")

    def __init__(self, block_idx, k_ptrs, v_ptrs):
        self.idx, self.k_ptrs, self.v_ptrs, self.ref_count = block_idx, k_ptrs, v_ptrs, 0

    def add_link(self):
        self.ref_count += 1

    def remove_link(self):
        self.ref_count -= 1

    def has_link(self):
        return self.ref_count > 0

    def is_shared(self):
        return self.ref_count > 1

    def get_k_ptr(self, idx):
        return self.k_ptrs[idx]

    def get_v_ptr(self, idx):
        return self.v_ptrs[idx]


class GenerationSequence:
    def __init__(self, seq_idx, batch_idx):
        self.seq_idx, self.batch_idx = seq_idx, batch_idx

    def get_batch_idx(self):
        return self.batch_idx

    def get_seq_idx(self):
        return self.seq_idx

    def __eq__(self, other):
        return (getattr(other, "seq_idx", None), getattr(other, "batch_idx", None)) == (self.seq_idx, self.batch_idx)

    def __hash__(self):
        return self.seq_idx


class BlocksManager:
    def __init__(self, memory_pools, blocks, max_blocks_per_seq=128, beam_width=1):
        self.memory_pools, self.blocks = memory_pools, blocks
        self.max_blocks_per_seq, self.beam_width = max_blocks_per_seq, beam_width
        self.pointer_array = None
        self.elts_per_blocks = [p.nelement() // (2 * blocks) for p in memory_pools]
        block_bytes = [e * _SIZEOF[p.dtype] for e, p in zip(self.elts_per_blocks, memory_pools)]
        bases = [p.data_ptr() for p in memory_pools]
        self.free_blocks = deque(
            Block(bi, [b + bi * bb for b, bb in zip(bases, block_bytes)],
                  [b + (blocks + bi) * bb for b, bb in zip(bases, block_bytes)]) for bi in range(blocks))
        self.allocated_blocks = {}  # owner -> [beam][block]

    def has_free_block(self):
        return len(self.free_blocks) > 0

    def _take(self):
        if not self.free_blocks:
            raise RuntimeError("Can't allocate new block for KV cache")
        return self.free_blocks.popleft()

    def allocate(self, owner, share_across_beam=False):
        beams = self.allocated_blocks.setdefault(owner, [[] for _ in range(self.beam_width)])
        block = None
        for bi in range(self.beam_width):
            if block is None or not share_across_beam:
                block = self._take()
            block.add_link()
            beams[bi].append(block)

    def replace_shared_block(self, owner, block_idx):
        beams = self.allocated_blocks[owner]
        if not beams[0][block_idx].is_shared():
            return
        for bi in range(self.beam_width):
            blk = beams[bi][block_idx]
            blk.remove_link()
            if not blk.has_link():
                self.free_blocks.append(blk)
        for bi in range(self.beam_width):
            blk = self._take()
            blk.add_link()
            beams[bi][block_idx] = blk

    def free(self, owner):
        for beam in self.allocated_blocks.pop(owner):
            for blk in beam:
                blk.remove_link()
                if not blk.has_link():
                    self.free_blocks.append(blk)

    def get_number_blocks(self, owner):
        return len(self.allocated_blocks[owner][0])

    def get_mempool_pointer(self, block_idx, pool, elts_per_block):
        return pool.data_ptr() + block_idx * elts_per_block * _SIZEOF[pool.dtype]

    def get_pointer_array(self, pool_idx, beam_width):
        """[batch, beam_width, 2, max_blocks_per_seq] int64 block addresses of pool `pool_idx` (0 where unallocated)."""
        assert beam_width <= self.beam_width
        arr = np.zeros((len(self.allocated_blocks), beam_width, 2, self.max_blocks_per_seq), dtype=np.int64)
        for owner, beams in self.allocated_blocks.items():
            b = owner.get_batch_idx()
            for bi in range(beam_width):
                for j, blk in enumerate(beams[bi]):
                    arr[b, bi, 0, j] = blk.get_k_ptr(pool_idx)
                    arr[b, bi, 1, j] = blk.get_v_ptr(pool_idx)
        self.pointer_array = torch.from_numpy(arr)
        return self.pointer_array

    def get_continuous_caches(self, pool_idx):
        """Debug view: every sequence's blocks concatenated, [batch, 2, max_blocks * elts_per_block]."""
        assert self.beam_width == 1
        epb = self.elts_per_blocks[pool_idx]
        pool = self.memory_pools[pool_idx].flatten()
        out = torch.zeros(len(self.allocated_blocks), 2, self.max_blocks_per_seq * epb, dtype=pool.dtype, device=pool.device)
        for owner, beams in self.allocated_blocks.items():
            for j, blk in enumerate(beams[0]):
                k0 = blk.idx * epb
                v0 = k0 + self.blocks * epb
                out[owner.get_batch_idx(), 0, j * epb:(j + 1) * epb] = pool[k0:k0 + epb]
                out[owner.get_batch_idx(), 1, j * epb:(j + 1) * epb] = pool[v0:v0 + epb]
        return out


class KVCacheManager:
    def __init__(self, memory_pools, blocks, tokens_per_block, max_blocks_per_seq, max_attention_window_size,
                 sink_token_len, beam_width=1, use_one_more_block=False):
        self.blocks_manager = BlocksManager(memory_pools, blocks, max_blocks_per_seq, beam_width)
        self.num_pools = len(memory_pools)
        self.tokens_per_block = tokens_per_block
        self.max_attention_window_size = max_attention_window_size
        self.sink_token_len = sink_token_len
        self.beam_width = beam_width
        # sink tokens do not share a block with the rest: a bubble pads them to a block boundary (:281-291)
        rem = sink_token_len % tokens_per_block
        self.bubble_len = 0 if rem == 0 else tokens_per_block - rem
        self.sink_block_token_num = sink_token_len + self.bubble_len
        self.max_token_num = max_attention_window_size + self.bubble_len + (tokens_per_block if use_one_more_block else 0)
        self.lens = []
        self.sequences = []

    def _next_index_in_cache(self, length):
        cyclic = self.max_token_num - self.sink_block_token_num
        return self.sink_block_token_num + (length - self.sink_block_token_num) % cyclic

    def step(self, finished):
        """One generation step: a new block where the next token starts one, finished sequences released (:297-340)."""
        cyclic = self.max_token_num - self.sink_block_token_num
        for seq in self.sequences:
            b = seq.get_batch_idx()
            nxt = self._next_index_in_cache(self.lens[b])
            if not finished[b] and (nxt % self.tokens_per_block == 0 or (nxt - self.sink_block_token_num) % cyclic == 0):
                if self.lens[b] < self.max_token_num:
                    self.blocks_manager.allocate(seq)
                elif self.beam_width > 1:
                    self.blocks_manager.replace_shared_block(seq, nxt // self.tokens_per_block)
            self.lens[b] += 1
        for seq, fin in zip(list(self.sequences), finished):
            if fin:
                self.blocks_manager.free(seq)
        self.lens = [l for l, f in zip(self.lens, finished) if not f]
        kept = [s for s, f in zip(self.sequences, finished) if not f]
        for i, s in enumerate(kept):
            s.batch_idx = i
        self.sequences = kept

    def add_sequence(self, sequence, context_len):
        """Register a sequence and allocate the blocks of its context (:342-371)."""
        seq_len = context_len + self.bubble_len
        self.lens.append(seq_len)
        self.sequences.append(sequence)
        final_idx = self._next_index_in_cache(seq_len - 1)
        unshared = final_idx // self.tokens_per_block if final_idx % self.tokens_per_block > 0 else -1
        seq_len = min(seq_len, self.max_token_num)
        for i in range(-(-seq_len // self.tokens_per_block)):
            self.blocks_manager.allocate(sequence, share_across_beam=i != unshared)

    def get_block_pointers(self, beam_width):
        """[pools, batch, beam_width, 2, max_blocks_per_seq] int64 (:373-384)."""
        return torch.stack([self.blocks_manager.get_pointer_array(p, beam_width) for p in range(self.num_pools)], dim=0)
