"""Recurrent containers (reference pfrl/nn/recurrent.py:1-30, recurrent_sequential.py:12-62,
recurrent_branched.py:6-26).

The calling convention is ``torch.nn.LSTM``'s with packed input:
``module(packed_sequences, recurrent_state) -> (packed_output, new_recurrent_state)``, where a
``None`` state means "start of sequence"."""
from torch import nn

from pfrl_amd.utils.recurrent import (get_packed_sequence_info, is_recurrent,
                                      unwrap_packed_sequences_recursive,
                                      wrap_packed_sequences_recursive)


class Recurrent(object):
    """Marker + interface of a stateful module."""

    def forward(self, packed_input, recurrent_state):
        raise NotImplementedError


class RecurrentSequential(Recurrent, nn.Sequential):
    """``nn.Sequential`` whose recurrent members (LSTM / GRU / RNN / ``Recurrent``) each own one
    slot of the tuple state.  Stateless members run once on the flat time-major tensor under the
    ``PackedSequence`` -- all time steps of all sequences in one launch -- and the packing is
    re-applied only where a recurrent member needs it."""

    @property
    def recurrent_children(self):
        return tuple(m for m in self if is_recurrent(m))

    def forward(self, sequences, recurrent_state):
        n_slots = len(self.recurrent_children)
        if recurrent_state is None:
            recurrent_state = (None,) * n_slots
        assert len(recurrent_state) == n_slots
        packing = get_packed_sequence_info(sequences)
        h, packed = sequences, True
        slot, new_state = 0, []
        for member in self:
            if is_recurrent(member):
                if not packed:
                    h, packed = wrap_packed_sequences_recursive(h, *packing), True
                h, s = member(h, recurrent_state[slot])
                new_state.append(s)
                slot += 1
            else:
                if packed:
                    h, packed = unwrap_packed_sequences_recursive(h), False
                h = member(h)
        if not packed:
            h = wrap_packed_sequences_recursive(h, *packing)
        assert slot == n_slots
        return h, tuple(new_state)


class RecurrentBranched(Recurrent, nn.ModuleList):
    """Feeds the same packed input to every branch; outputs and states are tuples in branch order."""

    def __init__(self, *modules):
        super().__init__(modules)

    def forward(self, sequences, recurrent_state):
        if recurrent_state is None:
            recurrent_state = (None,) * len(self)
        outs, states = [], []
        for branch, s in zip(self, recurrent_state):
            y, s = branch(sequences, s)
            outs.append(y)
            states.append(s)
        return tuple(outs), tuple(states)
