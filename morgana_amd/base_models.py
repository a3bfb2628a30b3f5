"""Mirror of the model contract in ``morgana.base_models``: ``BaseModel`` :9-253 and ``BaseSPSS`` :256-285.

A model written against the reference (subclass ``BaseSPSS``; implement ``normaliser_sources`` /
``train_data_sources`` / ``predict`` / ``loss``) runs unchanged on top of this class, with ``morgana_amd.utils`` /
``morgana_amd.losses`` in place of ``morgana.utils`` / ``morgana.losses``.
"""
import os

import torch
import torch.nn as nn

from . import metrics


class BaseModel(nn.Module):
    """Abstract model with the attributes ExperimentBuilder relies on (base_models.py:27-34)."""

    # Order of operations of the layers behind ``utils.upsample_to_repetitions`` (not in the reference): None = the process default
    # (MORGANA_PHONE_RATE), True = layers that commute with the repetition run once per phone row, False = every product on the
    # frame rows as the reference orders them.  Models hand it to ``upsample_to_repetitions(phone_rate=)``; same outputs either way.
    phone_rate = None

    def __init__(self):
        super(BaseModel, self).__init__()
        self.normalisers = {}
        self.mode = ''
        self.metrics = metrics.Handler(loss=metrics.Mean())
        self.step = 0
        self.tensorboard = None

    def finalise_init(self):
        pass

    def normaliser_sources(self):
        return {}

    def train_data_sources(self):
        raise NotImplementedError("Required for training.")

    def valid_data_sources(self):
        return self.train_data_sources()

    def test_data_sources(self):
        return self.valid_data_sources()

    def forward(self, features):
        raise NotImplementedError

    def predict(self, features):
        raise NotImplementedError

    def loss(self, features, output_features):
        raise NotImplementedError

    def save_parameters(self, experiment_dir, epoch):
        """``{experiment_dir}/checkpoints/epoch_{N}.pt`` holding ``state_dict()`` (base_models.py:142-154)."""
        path = os.path.join(experiment_dir, 'checkpoints', 'epoch_{}.pt'.format(epoch))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.state_dict(), path)
        return path

    def load_parameters(self, checkpoint_path, strict=True, device=None):
        """base_models.py:156-175."""
        device = device if device is not None else next(self.parameters()).device
        state = torch.load(checkpoint_path, map_location=device)
        self.load_state_dict(state, strict=strict)

    def step_input_keys(self, features):
        """Names of the tensors of ``features`` that the captured training step READS (``graphs.GraphedStepCache`` copies only those into a
        graph's static buffers when it replays the step on a new batch; the other entries are handed over by reference).  None = all of
        them (the default: correct for any model).  A model whose step reads an operand table instead of the fp32 feature it was made
        from names the table and leaves the feature out - the copy of the 49 MB ``normalised_lab`` is most of what a replay on a new
        batch would cost at BASELINE config C2."""
        return None

    def bf16_table_features(self):
        """Names of the phone-level input features whose bf16 operand table this model reads when it runs in bf16 precision (the
        loader then carries ``name + '__bf16_table'`` next to the feature: ``data.DeviceBatches.use_bf16_tables``); () otherwise."""
        return ()

    # analysis hooks (no-ops here; base_models.py:177-253)
    def analysis_for_train_batch(self, features, output_features, out_dir, **kwargs):
        pass

    def analysis_for_train_epoch(self, out_dir, **kwargs):
        pass

    def analysis_for_valid_batch(self, features, output_features, out_dir, **kwargs):
        pass

    def analysis_for_valid_epoch(self, out_dir, **kwargs):
        pass

    def analysis_for_test_batch(self, features, output_features, out_dir, **kwargs):
        self.analysis_for_valid_batch(features, output_features, out_dir, **kwargs)

    def analysis_for_test_epoch(self, out_dir, **kwargs):
        self.analysis_for_valid_epoch(out_dir, **kwargs)


class BaseSPSS(BaseModel):
    """Abstract SPSS acoustic model: ``forward = predict; loss`` (base_models.py:276-285)."""

    def __init__(self):
        super(BaseSPSS, self).__init__()

    def forward(self, features):
        output_features = self.predict(features)
        loss = self.loss(features, output_features)
        return loss, output_features
