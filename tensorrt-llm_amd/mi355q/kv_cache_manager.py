"""Paged-KV block bookkeeping: the producer of the `kv_cache_block_pointers` tensor GPTAttention consumes.

Same public classes and methods as tensorrt_llm/runtime/kv_cache_manager.py (Block, GenerationSequence, BlocksManager,
KVCacheManager: what a caller of the reference's runtime touches), same pool layout (a pool holds all K blocks, then all
V blocks), same hand-out order (first freed, first reused) and the same pointer tensor
`[pools, batch, beam, 2, max_blocks_per_seq]` -- but its own data structures: the state is a flat integer block table
`[sequence slot, beam, max_blocks_per_seq]` of block indices plus a reference-count vector and a ring of free indices,
all numpy arrays updated in place; the pointer tensor of a pool is one vectorised expression over that table
(`base + index * block_bytes`, V = K + blocks * block_bytes) instead of a nest of Python lists of block objects rebuilt
per step.  Host-only Python, like the reference's."""
import numpy as np
import torch

_SIZEOF = {torch.float32: 4, torch.float16: 2, torch.bfloat16: 2, torch.int8: 1, torch.uint8: 1}
if hasattr(torch, "float8_e4m3fn"):
    _SIZEOF[torch.float8_e4m3fn] = 1


class GenerationSequence:
    """A request in the batch: (sequence id, current row in the batch)."""

    def __init__(self, seq_idx, batch_idx):
        self.seq_idx, self.batch_idx = seq_idx, batch_idx

    def get_batch_idx(self):
        return self.batch_idx

    def get_seq_idx(self):
        return self.seq_idx

    def __eq__(self, other):
        return isinstance(other, GenerationSequence) and (other.seq_idx, other.batch_idx) == (self.seq_idx, self.batch_idx)

    def __hash__(self):
        return hash(self.seq_idx)


class Block:
    """A view of one cache block of a BlocksManager (index, per-pool K / V addresses, reference count)."""

    __slots__ = ("idx", "_mgr")

    def __init__(self, idx, mgr):
        self.idx, self._mgr = int(idx), mgr

    @property
    def ref_count(self):
        return int(self._mgr._refs[self.idx])

    def has_link(self):
        return self.ref_count > 0

    def is_shared(self):
        return self.ref_count > 1

    def get_k_ptr(self, pool_idx):
        return int(self._mgr._k_base[pool_idx] + self.idx * self._mgr._block_bytes[pool_idx])

    def get_v_ptr(self, pool_idx):
        return self.get_k_ptr(pool_idx) + self._mgr.blocks * int(self._mgr._block_bytes[pool_idx])

    def __eq__(self, other):
        return isinstance(other, Block) and other.idx == self.idx and other._mgr is self._mgr

    def __hash__(self):
        return self.idx


class BlocksManager:
    _NONE = -1

    def __init__(self, memory_pools, blocks, max_blocks_per_seq=128, beam_width=1):
        self.memory_pools, self.blocks = memory_pools, blocks
        self.max_blocks_per_seq, self.beam_width = max_blocks_per_seq, beam_width
        self.elts_per_blocks = [p.nelement() // (2 * blocks) for p in memory_pools]
        self._block_bytes = np.array([e * _SIZEOF[p.dtype] for e, p in zip(self.elts_per_blocks, memory_pools)], np.int64)
        self._k_base = np.array([p.data_ptr() for p in memory_pools], np.int64)
        self._refs = np.zeros(blocks, np.int32)
        # ring of free block indices: take at _head, give back at (_head + _nfree) % blocks
        self._ring = np.arange(blocks, dtype=np.int32)
        self._head, self._nfree = 0, blocks
        # block table: one slot per live sequence
        self._table = np.full((4, beam_width, max_blocks_per_seq), self._NONE, np.int32)
        self._count = np.zeros(4, np.int32)
        self._slot_of = {}            # GenerationSequence -> slot (insertion order = admission order)
        self._free_slots = [3, 2, 1, 0]
        self.pointer_array = None

    # -- free ring
    def has_free_block(self):
        return self._nfree > 0

    def _take(self):
        if self._nfree == 0:
            raise RuntimeError("Can't allocate new block for KV cache")
        idx = int(self._ring[self._head])
        self._head = (self._head + 1) % self.blocks
        self._nfree -= 1
        return idx

    def _release(self, indices):
        """Drop one reference from each index (an index may repeat: blocks shared across beams)."""
        for idx in indices:
            self._refs[idx] -= 1
            if self._refs[idx] == 0:
                self._ring[(self._head + self._nfree) % self.blocks] = idx
                self._nfree += 1

    @property
    def free_blocks(self):
        """The free blocks in hand-out order (views; for inspection)."""
        return [Block(self._ring[(self._head + i) % self.blocks], self) for i in range(self._nfree)]

    @property
    def allocated_blocks(self):
        """{sequence: [beam][block] of Block views} (for inspection; the state itself is the integer table)."""
        return {seq: [[Block(i, self) for i in self._table[slot, b, :self._count[slot]]] for b in range(self.beam_width)]
                for seq, slot in self._slot_of.items()}

    # -- sequences
    def _slot(self, owner, create=False):
        slot = self._slot_of.get(owner)
        if slot is None:
            if not create:
                raise KeyError(owner)
            if not self._free_slots:
                old = self._table.shape[0]
                self._table = np.concatenate([self._table, np.full_like(self._table, self._NONE)], 0)
                self._count = np.concatenate([self._count, np.zeros(old, np.int32)])
                self._free_slots = list(range(2 * old - 1, old - 1, -1))
            slot = self._free_slots.pop()
            self._slot_of[owner] = slot
        return slot

    def allocate(self, owner, share_across_beam=False):
        """One more block for `owner`: a block of its own per beam, or one block every beam points at."""
        slot = self._slot(owner, create=True)
        j = int(self._count[slot])
        assert j < self.max_blocks_per_seq, "sequence exceeds max_blocks_per_seq"
        if share_across_beam:
            idx = self._take()
            self._refs[idx] += self.beam_width
            self._table[slot, :, j] = idx
        else:
            if self._nfree < self.beam_width:
                raise RuntimeError("Can't allocate new block for KV cache")
            for b in range(self.beam_width):
                idx = self._take()
                self._refs[idx] += 1
                self._table[slot, b, j] = idx
        self._count[slot] = j + 1

    def replace_shared_block(self, owner, block_idx):
        """Beam search writes into block `block_idx`: if the beams still share it, give each its own."""
        slot = self._slot(owner)
        col = self._table[slot, :, block_idx]
        if self._refs[col[0]] <= 1:
            return
        self._release(col.tolist())
        for b in range(self.beam_width):
            idx = self._take()
            self._refs[idx] += 1
            col[b] = idx

    def free(self, owner):
        slot = self._slot_of.pop(owner)
        n = int(self._count[slot])
        self._release(self._table[slot, :, :n].reshape(-1).tolist())   # beam-major, like walking beam by beam
        self._table[slot] = self._NONE
        self._count[slot] = 0
        self._free_slots.append(slot)

    def get_number_blocks(self, owner):
        return int(self._count[self._slot(owner)])

    def get_mempool_pointer(self, block_idx, pool, elts_per_block):
        return pool.data_ptr() + block_idx * elts_per_block * _SIZEOF[pool.dtype]

    # -- what the attention plugin consumes
    def get_pointer_array(self, pool_idx, beam_width):
        """[batch, beam_width, 2, max_blocks_per_seq] int64 block addresses of pool `pool_idx` (0 where unallocated);
        row = the sequence's batch index."""
        assert beam_width <= self.beam_width
        seqs = list(self._slot_of)
        arr = np.zeros((len(seqs), beam_width, 2, self.max_blocks_per_seq), np.int64)
        if seqs:
            rows = np.array([s.get_batch_idx() for s in seqs])
            idx = self._table[[self._slot_of[s] for s in seqs], :beam_width].astype(np.int64)     # [n, beam, max_blocks]
            live = idx >= 0
            bb, base = int(self._block_bytes[pool_idx]), int(self._k_base[pool_idx])
            arr[rows, :, 0] = np.where(live, base + idx * bb, 0)
            arr[rows, :, 1] = np.where(live, base + (idx + self.blocks) * bb, 0)
        self.pointer_array = torch.from_numpy(arr)
        return self.pointer_array

    def get_continuous_caches(self, pool_idx):
        """Debug view: every sequence's blocks concatenated, [batch, 2, max_blocks_per_seq * elts_per_block]."""
        assert self.beam_width == 1
        epb = self.elts_per_blocks[pool_idx]
        pool = self.memory_pools[pool_idx].reshape(2, self.blocks, epb)   # K half, V half
        out = torch.zeros(len(self._slot_of), 2, self.max_blocks_per_seq, epb, dtype=pool.dtype, device=pool.device)
        for seq, slot in self._slot_of.items():
            n = int(self._count[slot])
            ids = torch.from_numpy(self._table[slot, 0, :n].astype(np.int64)).to(pool.device)
            out[seq.get_batch_idx(), :, :n] = pool[:, ids]
        return out.reshape(len(self._slot_of), 2, -1)


class KVCacheManager:
    """Sequence lengths -> block demand, for a (possibly cyclic, possibly sink-token) attention window."""

    def __init__(self, memory_pools, blocks, tokens_per_block, max_blocks_per_seq, max_attention_window_size,
                 sink_token_len, beam_width=1, use_one_more_block=False):
        self.blocks_manager = BlocksManager(memory_pools, blocks, max_blocks_per_seq, beam_width)
        self.num_pools = len(memory_pools)
        self.tokens_per_block = tokens_per_block
        self.max_attention_window_size = max_attention_window_size
        self.sink_token_len = sink_token_len
        self.beam_width = beam_width
        # sink tokens never share a block with the window: a bubble pads them up to a block boundary
        self.bubble_len = -sink_token_len % tokens_per_block
        self.sink_block_token_num = sink_token_len + self.bubble_len
        self.max_token_num = max_attention_window_size + self.bubble_len + (tokens_per_block if use_one_more_block else 0)
        self.lens = []
        self.sequences = []

    def _slot_in_window(self, length):
        """Cache position the token with index `length` lands on (the window behind the sink blocks is cyclic)."""
        window = self.max_token_num - self.sink_block_token_num
        return self.sink_block_token_num + (length - self.sink_block_token_num) % window

    def add_sequence(self, sequence, context_len):
        """Admit a sequence and give it the blocks of its context.  With beams the blocks are shared, except the one
        the next token will be written into when it is part-filled."""
        length = context_len + self.bubble_len
        self.lens.append(length)
        self.sequences.append(sequence)
        last = self._slot_in_window(length - 1)
        private = last // self.tokens_per_block if last % self.tokens_per_block else -1
        for i in range(-(-min(length, self.max_token_num) // self.tokens_per_block)):
            self.blocks_manager.allocate(sequence, share_across_beam=(i != private))

    def step(self, finished):
        """One generation step: sequences whose next token opens a block get one (while the window still grows; once it
        is full, beams un-share the block they are about to overwrite), finished sequences give theirs back and the
        remaining ones close ranks in the batch."""
        window = self.max_token_num - self.sink_block_token_num
        for seq in self.sequences:
            b = seq.get_batch_idx()
            pos = self._slot_in_window(self.lens[b])
            opens_block = pos % self.tokens_per_block == 0 or (pos - self.sink_block_token_num) % window == 0
            if opens_block and not finished[b]:
                if self.lens[b] < self.max_token_num:
                    self.blocks_manager.allocate(seq)
                elif self.beam_width > 1:
                    self.blocks_manager.replace_shared_block(seq, pos // self.tokens_per_block)
            self.lens[b] += 1
        survivors = []
        for seq in self.sequences:
            if finished[seq.get_batch_idx()]:
                self.blocks_manager.free(seq)
            else:
                survivors.append(seq)
        self.lens = [self.lens[s.get_batch_idx()] for s in survivors]
        for row, s in enumerate(survivors):
            # the sequence object is the key of the block table: re-key it under its new batch row
            slot = self.blocks_manager._slot_of.pop(s)
            s.batch_idx = row
            self.blocks_manager._slot_of[s] = slot
        self.sequences = survivors

    def get_block_pointers(self, beam_width):
        """[pools, batch, beam_width, 2, max_blocks_per_seq] int64."""
        return torch.stack([self.blocks_manager.get_pointer_array(p, beam_width) for p in range(self.num_pools)], dim=0)
