"""Training loop with the reference's control flow (``training_utils.py:8-97``): fixed-chunk or random minibatches,
validation every ``test_freq`` iterations, best-on-disk checkpoint, early stopping, reload of the best checkpoint.

Added for multi-GPU: when ``torch.distributed`` is initialised each rank trains on its own shard of every minibatch
and the flat gradient bucket is averaged with one RCCL all-reduce per step (``hode.parallel.GradBucket``).
"""
import time

import torch

from hode.parallel import GradBucket, is_distributed, rank0_print


def _trainable(optimizer):
    return [p for g in optimizer.param_groups for p in g["params"]]


def variational_training_loop(niters, data_generator, model, batch_size, optimizer, test_freq, best_on_disk=1e9,
                              early_stop=5, path="model/", shuffle=True, train_fold="train"):
    best_loss = 1e9
    stale = 0
    fold_size = data_generator.train_size if train_fold == "train" else data_generator.val_size
    train_chunk = fold_size // batch_size
    bucket = GradBucket(_trainable(optimizer)) if is_distributed() else None

    start = time.time()
    for itr in range(1, niters + 1):
        if shuffle:
            data = data_generator.get_mini_batch(train_fold, batch_size)
        else:
            data = data_generator.get_split(train_fold, batch_size, itr % train_chunk)
        optimizer.zero_grad()
        try:
            loss = model.loss(data)
        except RuntimeError as e:  # solver blow-up (non-finite state, dt underflow) ends this restart
            rank0_print(e)
            break
        loss.backward()
        if bucket is not None:
            bucket.all_reduce_mean()
        optimizer.step()

        if itr % test_freq == 0:
            with torch.no_grad():
                total = 0
                for chunk in range(data_generator.val_size // batch_size):
                    data = data_generator.get_split("val", batch_size, chunk)
                    try:
                        total += model.loss(data).item()
                    except RuntimeError as e:
                        total += 1e9
                        rank0_print(e)
                        break
                rank0_print("Iter {:04d} | Total Loss {:.6f} | Train Loss {:.6f}".format(itr, total, loss.item()))
                if total < best_loss:
                    best_loss, stale = total, 0
                else:
                    stale += 1
                if total < best_on_disk:
                    best_on_disk = total
                    model.save(path, itr, best_on_disk)
        if stale >= early_stop:
            break
    end = time.time()

    try:
        best = torch.load(path + model.model_name)
    except FileNotFoundError:
        model.save(path, 0, best_on_disk)
        best = torch.load(path + model.model_name)
    model.encoder.load_state_dict(best["encoder_state_dict"])
    model.decoder.load_state_dict(best["decoder_state_dict"])
    best_loss = best["best_loss"]
    rank0_print("Time: {}".format(end - start))
    rank0_print("Overall best loss: {:.6f}".format(best_loss))
    return model, best_loss, end - start
