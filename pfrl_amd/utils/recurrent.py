"""Recurrent-state and packed-sequence helpers (reference pfrl/utils/recurrent.py:8-357).

A *recurrent state* is a pytree: ``None`` (not yet initialised), a tensor / ndarray laid out
``(layers, batch, hidden)`` -- batch on axis 1, as ``torch.nn.LSTM`` keeps it -- or a tuple of
recurrent states.  Packed data is either a ``PackedSequence``, a tuple of them, or something else
that is passed through.  Every function here is a map or a zip over one of those two trees, so the
tree walking is factored into ``_map_state`` / ``_map_packed`` and the public functions only say
what happens at a leaf.
"""
import numpy as np
import torch
from torch import nn
from torch.nn.utils.rnn import PackedSequence, pack_sequence

_RNN_TYPES = (nn.LSTM, nn.RNN, nn.GRU)


def is_recurrent(layer):
    """True iff ``layer`` takes and returns a recurrent state (reference :8-28)."""
    from pfrl_amd.nn.recurrent import Recurrent

    return isinstance(layer, _RNN_TYPES + (Recurrent,))


def _map_state(leaf_fn, recurrent_state, leaf_types=(torch.Tensor,)):
    if recurrent_state is None:
        return None
    if isinstance(recurrent_state, leaf_types):
        return leaf_fn(recurrent_state)
    if isinstance(recurrent_state, tuple):
        return tuple(_map_state(leaf_fn, s, leaf_types) for s in recurrent_state)
    raise ValueError("Invalid recurrent state: {}".format(recurrent_state))


def mask_recurrent_state_at(recurrent_state, indices):
    """A copy with the batch entries ``indices`` zeroed: those sequences restart (reference :31-53)."""

    def zero_columns(s):
        keep = torch.ones_like(s)
        keep[:, indices] = 0
        return s * keep

    return _map_state(zero_columns, recurrent_state)


def get_recurrent_state_at(recurrent_state, indices, detach):
    """The state of the batch entries ``indices`` (batch axis dropped for an int; reference :56-79)."""
    return _map_state(lambda s: (s.detach() if detach else s)[:, indices], recurrent_state)


def concatenate_recurrent_states(split_recurrent_states):
    """Stack per-sequence states (from ``get_recurrent_state_at(..., int)``) along a new batch
    axis 1; ``None`` members count as zeros shaped like the others (reference :82-118)."""
    template = next((s for s in split_recurrent_states if s is not None), None)
    if template is None:
        return None
    if isinstance(template, tuple):
        return tuple(
            concatenate_recurrent_states([None if s is None else s[i] for s in split_recurrent_states])
            for i in range(len(template)))
    if isinstance(template, torch.Tensor):
        zeros, stack = torch.zeros_like, lambda xs: torch.stack(xs, dim=1)
    elif isinstance(template, np.ndarray):
        zeros, stack = np.zeros_like, lambda xs: np.stack(xs, axis=1)
    else:
        raise ValueError("Invalid recurrent state: {}".format(template))
    return stack([zeros(template) if s is None else s for s in split_recurrent_states])


def recurrent_state_as_numpy(recurrent_state):
    """Tensors -> ndarrays, for storage in a replay buffer (reference :302-318)."""
    return _map_state(lambda s: s.detach().cpu().numpy(), recurrent_state)


def recurrent_state_from_numpy(recurrent_state, device):
    """ndarrays -> tensors on ``device`` (reference :321-338)."""
    return _map_state(lambda a: torch.from_numpy(a).to(device), recurrent_state,
                      leaf_types=(np.ndarray,))


def detach_recurrent_state(recurrent_state):
    """Cut the graph at the state, e.g. between PPO rollouts (reference :341-357)."""
    return _map_state(torch.Tensor.detach, recurrent_state)


# ---- packed sequences -------------------------------------------------------------------------
def _map_packed(packed_fn, x):
    if isinstance(x, PackedSequence):
        return packed_fn(x)
    if isinstance(x, tuple):
        return tuple(_map_packed(packed_fn, y) for y in x)
    return x


def unwrap_packed_sequences_recursive(packed):
    """The time-major flat tensor(s) under the ``PackedSequence``(s), unpadded (reference :220-246)."""
    return _map_packed(lambda p: p.data, packed)


unpack_sequences_as_one_step_batch = unwrap_packed_sequences_recursive  # reference :134-140


def wrap_packed_sequences_recursive(unwrapped, batch_sizes, sorted_indices):
    """Inverse of ``unwrap_packed_sequences_recursive`` given the packing info (reference :195-217)."""
    if isinstance(unwrapped, torch.Tensor):
        return PackedSequence(unwrapped, batch_sizes=batch_sizes, sorted_indices=sorted_indices)
    if isinstance(unwrapped, tuple):
        return tuple(wrap_packed_sequences_recursive(u, batch_sizes, sorted_indices)
                     for u in unwrapped)
    return unwrapped


def get_packed_sequence_info(packed):
    """``(batch_sizes, sorted_indices)`` of the first ``PackedSequence`` found (reference :280-299)."""
    if isinstance(packed, PackedSequence):
        return packed.batch_sizes, packed.sorted_indices
    if isinstance(packed, tuple):
        for member in packed:
            info = get_packed_sequence_info(member)
            if info is not None:
                return info
    return None


def pack_sequences_recursive(sequences):
    """``pack_sequence`` that also handles sequences of tuples of tensors: the i-th members are
    packed together, giving a tuple of ``PackedSequence`` (reference :249-277)."""
    assert sequences
    head = sequences[0]
    if isinstance(head, torch.Tensor):
        return pack_sequence(sequences)
    if isinstance(head, tuple):
        return tuple(pack_sequences_recursive([seq[i] for seq in sequences])
                     for i in range(len(head)))
    return sequences


def pack_one_step_batch_as_sequences(xs):
    """A ``(B, ...)`` batch as B sequences of length one (reference :121-131)."""
    if isinstance(xs, tuple):
        return tuple(pack_one_step_batch_as_sequences(x) for x in xs)
    assert isinstance(xs, torch.Tensor)
    return pack_sequence(list(xs.split(1)))


def one_step_forward(rnn, batch_input, recurrent_state):
    """One batched step of a recurrent module: ``(output batch, new state)`` (reference :143-157)."""
    y, recurrent_state = rnn(pack_one_step_batch_as_sequences(batch_input), recurrent_state)
    return unwrap_packed_sequences_recursive(y), recurrent_state


def pack_and_forward(rnn, sequences, recurrent_state):
    """Whole-sequence forward; the output is the time-major flat tensor(s) (reference :160-174)."""
    y, recurrent_state = rnn(pack_sequences_recursive(sequences), recurrent_state)
    return unwrap_packed_sequences_recursive(y), recurrent_state


def flatten_sequences_time_first(sequences):
    """Items of batch-major ``sequences`` in the order ``pack_sequence`` lays a length-sorted
    batch out: all first items, then all second items of the sequences that have one, ...
    (reference :177-192)."""
    flat = []
    t = 0
    while True:
        column = [seq[t] for seq in sequences if len(seq) > t]
        if not column:
            return flat
        flat.extend(column)
        t += 1
