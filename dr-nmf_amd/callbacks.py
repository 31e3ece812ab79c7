"""The slice of the Keras callback protocol the reference's training loop uses
(enhance.py:1134-1157): `fit(..., callbacks=[LossHistory(histfile), ModelCheckpoint(savefile,
save_best_only=True, save_weights_only=True), EarlyStopping('val_loss', patience)])`.
[K2.0.4-memory] for the semantics of monitor / patience / mode='auto'."""
import pickle

import numpy as np


class Callback(object):
    """keras.callbacks.Callback: hooks are optional; `self.model` is set by fit()."""
    model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


def _better(monitor, mode):
    if mode == 'max' or (mode == 'auto' and 'acc' in monitor):
        return (lambda a, b: a > b), -np.inf
    return (lambda a, b: a < b), np.inf


class EarlyStopping(Callback):
    """Stop when `monitor` has not improved (by more than min_delta) for more than `patience`
    epochs."""

    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto'):
        self.monitor, self.min_delta, self.patience, self.verbose = (monitor, abs(min_delta),
                                                                       patience, verbose)
        self.is_better, self.best = _better(monitor, mode)
        self.sign = -1.0 if self.best == np.inf else 1.0
        self.wait, self.stopped_epoch = 0, 0

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch = 0, 0
        self.best = np.inf if self.sign < 0 else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        # Keras 2.0.4 negates min_delta for 'min' monitors and tests current - min_delta: an
        # improvement must exceed min_delta
        if self.is_better(cur - self.sign * self.min_delta, self.best):
            self.best, self.wait = cur, 0
        else:
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True
                if self.verbose:
                    print('Epoch %05d: early stopping' % epoch)
            self.wait += 1


class ModelCheckpoint(Callback):
    """Save the weights after every epoch, or only when `monitor` improves (save_best_only).
    The file is what `model.save_weights(filepath)` writes ('.npz' tree, or Keras HDF5 when h5py
    is available); `filepath` may contain '{epoch}' / '{val_loss}'-style fields."""

    rank0_only = True        # data parallelism: one writer (fit() drops it on the other ranks)

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False,
                 save_weights_only=False, mode='auto', period=1):
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.period = save_best_only, period
        self.save_weights_only = save_weights_only     # (only weights can be saved here)
        self.is_better, self.best = _better(monitor, mode)
        self.epochs_since_last_save = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        path = self.filepath.format(epoch=epoch, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self.is_better(cur, self.best):
                return
            if self.verbose:
                print('Epoch %05d: %s improved from %0.5f to %0.5f, saving model to %s'
                      % (epoch, self.monitor, self.best, cur, path))
            self.best = cur
        self.model.save_weights(path)


class LossHistory(Callback):
    """custom_callbacks.py:4-27: per-batch and per-epoch metric lists, pickled to `histfile` after
    every epoch as {'on_batch_end': {...}, 'on_epoch_end': {...}}."""

    rank0_only = True

    def __init__(self, histfile):
        self.histfile = histfile

    def on_train_begin(self, logs=None):
        self.metrics_on_batch_end = {}
        self.metrics_on_epoch_end = {}

    def on_batch_end(self, batch, logs=None):
        for key, val in (logs or {}).items():
            self.metrics_on_batch_end.setdefault(key, []).append(val)

    def on_epoch_end(self, epoch, logs=None):
        for key, val in (logs or {}).items():
            self.metrics_on_epoch_end.setdefault(key, []).append(val)
        with open(self.histfile, 'wb') as f:
            pickle.dump({'on_batch_end': self.metrics_on_batch_end,
                         'on_epoch_end': self.metrics_on_epoch_end}, f)
