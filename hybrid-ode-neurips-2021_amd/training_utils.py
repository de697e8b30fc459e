"""Training loop with the reference's control flow (``training_utils.py:8-97``): fixed-chunk or random minibatches,
validation every ``test_freq`` iterations, best-on-disk checkpoint, early stopping, reload of the best checkpoint.

Added for multi-GPU: when ``torch.distributed`` is initialised every rank draws the SAME minibatch (the
host generators are seeded identically -- this holds for ``DataGeneratorRoche`` and ``DeviceFolds(index_rng="host")``; with
``index_rng="device"`` it holds only if the ranks seed their device generators identically, otherwise the ranks train on
independent minibatches, which is still a valid data-parallel step), keeps its contiguous shard of the patients
(``hode.parallel.shard_batch``) and the flat gradient bucket is combined with one RCCL all-reduce per step
(``hode.parallel.GradBucket``), each rank weighted by its share of the patients (ragged shards included).  The ranks also
share torch's seed, so ``reparameterize`` draws the same eps block on every rank for DIFFERENT patients -- harmless (the
draws are i.i.d. per patient either way); seed the ranks apart (``torch.manual_seed(seed + rank)`` after the data
generators are built) if independent noise is wanted.  Every decision that
ends or redirects the loop is taken on all-reduced quantities, so the ranks leave it together: a solver failure on one
rank's shard travels in the bucket's flag slot and stops every rank before the optimiser step; the validation total is
the mean of the per-rank totals (losses are normalised per local batch, reference model.py:1179,1188); rank 0 alone
writes checkpoints and the others wait for it before reading.
"""
import time

import torch
import torch.distributed as dist

from hode.parallel import GradBucket, is_distributed, rank0_print, shard_batch


def _trainable(optimizer):
    return [p for g in optimizer.param_groups for p in g["params"]]


def _local(data):
    """This rank's contiguous shard of the batch and its share n_local / n_global of the patients.  Losses are normalised
    per LOCAL batch (reference model.py:1179,1188), so the global-batch loss / gradient is the share-weighted sum of the
    per-rank ones; with equal shards that is the plain mean, with ragged ones (batch_size % world != 0) it is not."""
    if not is_distributed():
        return data, 1.0
    n = next(iter(data.values())).shape[1]
    shard = shard_batch(data)
    return shard, next(iter(shard.values())).shape[1] / float(n)


def _validation_total(model, data_generator, batch_size):
    """Sum of the validation losses over the fold's chunks; a failing chunk counts 1e9 and ends the pass
    (reference :57-66).  Distributed: mean over ranks, and one rank's failure is every rank's."""
    total, failed = 0.0, 0.0
    for chunk in range(data_generator.val_size // batch_size):
        data, share = _local(data_generator.get_split("val", batch_size, chunk))
        try:
            total += model.loss(data).item() * share
        except RuntimeError as e:
            failed = 1.0
            print(e)
            break
    if is_distributed():
        dev = next(model.encoder.parameters()).device
        buf = torch.tensor([total, failed], dtype=torch.float64, device=dev)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        total, failed = float(buf[0]), float(buf[1])  # shares sum to 1 over the ranks: the sum IS the global-batch loss
    return total + (1e9 if failed else 0.0)


def variational_training_loop(niters, data_generator, model, batch_size, optimizer, test_freq, best_on_disk=1e9,
                              early_stop=5, path="model/", shuffle=True, train_fold="train"):
    best_loss = 1e9
    stale = 0
    fold_size = data_generator.train_size if train_fold == "train" else data_generator.val_size
    train_chunk = fold_size // batch_size
    distributed = is_distributed()
    bucket = GradBucket(_trainable(optimizer)) if distributed else None
    rank = dist.get_rank() if distributed else 0

    start = time.time()
    for itr in range(1, niters + 1):
        if shuffle:
            data = data_generator.get_mini_batch(train_fold, batch_size)
        else:
            data = data_generator.get_split(train_fold, batch_size, itr % train_chunk)
        data, share = _local(data)
        optimizer.zero_grad()
        failure = None
        try:
            loss = model.loss(data)
        except RuntimeError as e:  # solver blow-up (non-finite state, dt underflow) ends this restart
            failure = e
        if not distributed:
            if failure is not None:
                print(failure)
                break
            loss.backward()
        else:
            # the exchange is collective: a rank whose shard failed still takes part, with zero gradients and its flag up
            if failure is None:
                loss.backward()
            else:
                print(failure)
                optimizer.zero_grad()
            if bucket.all_reduce_mean(weight=share * dist.get_world_size(), failed=failure is not None):
                break
        optimizer.step()

        if itr % test_freq == 0:
            with torch.no_grad():
                total = _validation_total(model, data_generator, batch_size)
                rank0_print("Iter {:04d} | Total Loss {:.6f} | Train Loss {:.6f}".format(itr, total, loss.item()))
                if total < best_loss:
                    best_loss, stale = total, 0
                else:
                    stale += 1
                if total < best_on_disk:
                    best_on_disk = total
                    if rank == 0:
                        model.save(path, itr, best_on_disk)
        if stale >= early_stop:
            break
    end = time.time()

    if distributed:
        dist.barrier()  # rank 0's last checkpoint is on disk before anyone reads it
    try:
        best = torch.load(path + model.model_name)
    except FileNotFoundError:
        if rank == 0:
            model.save(path, 0, best_on_disk)
        if distributed:
            dist.barrier()
        best = torch.load(path + model.model_name)
    model.encoder.load_state_dict(best["encoder_state_dict"])
    model.decoder.load_state_dict(best["decoder_state_dict"])
    best_loss = best["best_loss"]
    rank0_print("Time: {}".format(end - start))
    rank0_print("Overall best loss: {:.6f}".format(best_loss))
    return model, best_loss, end - start


# ---------------------------------------------------------------------------------------------------------------------
# Evaluation (reference training_utils.py:100-279, :568-577).  Same signatures, prints and return values.  What changed:
# the mc_itr posterior draws are integrated in ONE decoder call over a batch of mc_itr * B latents (the reference loops
# mc_itr decoder calls) and scored by one `hode_ensemble_crps` launch that applies the linear readout on the fly (the
# reference stacks (T', B, obs, mc_itr) and calls properscoring.crps_ensemble per element from three nested Python
# loops).  The draws themselves are taken exactly as the reference takes them -- mc_itr consecutive
# `encoder.reparameterize(*encoder_out)` calls -- so a seeded run sees the same random numbers.
# ---------------------------------------------------------------------------------------------------------------------
import numpy as np

from hode import crps as _crps_mod

_ensemble_crps = _crps_mod.ensemble_crps  # tests swap in the CPU oracle here; the product path never does


def bootstrap_RMSE(err_sq):
    if type(err_sq) == np.ndarray:
        err_sq = torch.tensor(err_sq)
    rmse_list = []
    for _ in range(500):
        new_err = err_sq[torch.randint(len(err_sq), err_sq.shape, device=err_sq.device)]
        rmse_list.append(torch.sqrt(torch.mean(new_err)).item())
    return np.std(np.array(rmse_list))


def _posterior_scores(model, data, t0, mc_itr, real, expert_dim):
    """One test chunk: point-estimate errors and ensemble CRPS.  Returns
    (se_z0 (B,), sse_x (T', B), n_x (T', B), crps_z0 (B,) mean over expert dims, crps_x (T', B) mean over obs)."""
    x, a, mask = data["measurements"][:t0], data["actions"][:t0], data["masks"][:t0]
    z0 = data["latents"][0]
    if real:
        s = data["statics"][:t0]
        encoder_out = model.encoder(x, torch.cat([a, s], dim=-1), mask)
        z0_hat = encoder_out[0]
        x_hat, _ = model.decoder(z0_hat, data["actions"], data["statics"])
    else:
        encoder_out = model.encoder(x, a, mask)
        z0_hat = encoder_out[0]
        x_hat, _ = model.decoder(z0_hat, data["actions"])
    x_hat = x_hat[t0:, ...]
    se_z0 = torch.sum((z0[:, :expert_dim] - z0_hat[:, :expert_dim]) ** 2, dim=1)
    x_test, mask_test = data["measurements"][t0:], data["masks"][t0:]
    sse_x = torch.sum((x_test - x_hat) ** 2 * mask_test, dim=2)
    n_x = torch.sum(mask_test, dim=2)

    # ---- posterior ensemble: mc_itr draws -> one batch of mc_itr * B latents (member-major)
    M = int(mc_itr)
    B, D = z0_hat.shape
    z = torch.stack([model.encoder.reparameterize(*encoder_out) for _ in range(M)], dim=0)  # (M, B, D)
    z_flat = z.reshape(M * B, D)
    act = data["actions"].repeat(1, M, 1)
    if real:
        x_mc, _ = model.decoder(z_flat, act, data["statics"].repeat(1, M, 1))               # (T, M*B, obs)
        crps_x = _ensemble_crps(x_mc[t0:].contiguous(), x_test, M)
    else:
        lin = model.decoder.output_function[0]
        h_mc = model.decoder.latent(z_flat, act)                                            # (T, M*B, D)
        crps_x = _ensemble_crps(h_mc[t0:], x_test, M, weight=lin.weight, bias=lin.bias)
    crps_x = crps_x / x_test.shape[2]
    crps_z0 = _ensemble_crps(z_flat.unsqueeze(0), z0[:, :expert_dim].unsqueeze(0).contiguous(), M)[0] / expert_dim
    return se_z0, sse_x, n_x, crps_z0, crps_x


def _rmse_with_bootstrap(sq_err, resample_observed_only=True):
    """sqrt(mean) of a 1-D tensor of squared errors over its non-NaN entries (NaN = nothing observed) and the bootstrap
    standard deviation of that estimate (``bootstrap_RMSE``).  ``evaluate`` resamples the observed entries only
    (reference :186-189); ``evaluate_horizon`` hands the raw row to the bootstrap (reference :272), so a forecast step
    with an unobserved patient reports a NaN spread there, as the reference does."""
    seen = sq_err[~torch.isnan(sq_err)]
    return torch.sqrt(seen.mean()).item(), bootstrap_RMSE(seen if resample_observed_only else sq_err)


def _mean_with_standard_error(values, axis=None):
    """mean and standard error of the mean (population std / sqrt(count)) along ``axis``."""
    count = values.size if axis is None else values.shape[axis]
    return np.mean(values, axis=axis), np.std(values, axis=axis) / np.sqrt(count)


def evaluate(model, data_generator, batch_size, t0, mc_itr=50, real=False):
    """Reference ``training_utils.evaluate`` (:100-201): per test chunk the point-estimate errors and the ensemble CRPS
    of ``mc_itr`` posterior draws; prints the four ``name,value,sd`` lines and returns
    ``(rmse_z0, rmse_z0_sd, cprs_z0, rmse_x, rmse_x_sd, cprs_x)``."""
    per_chunk = {"se_z0": [], "mse_x": [], "crps_z0": [], "crps_x": []}
    with torch.no_grad():
        for chunk in range(data_generator.test_size // batch_size):
            data = data_generator.get_split("test", batch_size, chunk)
            se_z0, sse_x, n_x, crps_z0, crps_x = _posterior_scores(model, data, t0, mc_itr, real, data_generator.expert_dim)
            per_chunk["se_z0"].append(se_z0.cpu())
            per_chunk["mse_x"].append((sse_x.sum(dim=0) / n_x.sum(dim=0)).cpu())  # per patient, over the horizon
            per_chunk["crps_z0"].append(crps_z0.cpu().numpy())
            per_chunk["crps_x"].append(crps_x.mean(dim=0).cpu().numpy())
    rmse_z0, rmse_z0_sd = _rmse_with_bootstrap(torch.cat(per_chunk["se_z0"]))
    rmse_x, rmse_x_sd = _rmse_with_bootstrap(torch.cat(per_chunk["mse_x"]))
    cprs_z0, cprs_z0_sd = _mean_with_standard_error(np.concatenate(per_chunk["crps_z0"]))
    cprs_x, cprs_x_sd = _mean_with_standard_error(np.concatenate(per_chunk["crps_x"]))
    for name, value, sd in (("rmse_z0", rmse_z0, rmse_z0_sd), ("rmse_x", rmse_x, rmse_x_sd),
                            ("cprs_z0", cprs_z0, cprs_z0_sd), ("cprs_x", cprs_x, cprs_x_sd)):
        print("{},{:.4f},{:.4f}".format(name, value, sd))
    return rmse_z0, rmse_z0_sd, cprs_z0, rmse_x, rmse_x_sd, cprs_x


def evaluate_horizon(model, data_generator, batch_size, t0, mc_itr=10, real=False):
    """Reference ``training_utils.evaluate_horizon`` (:204-279): the same scores resolved per forecast step."""
    mse_x, crps_x_all = [], []
    with torch.no_grad():
        for chunk in range(data_generator.test_size // batch_size):
            data = data_generator.get_split("test", batch_size, chunk)
            _, sse_x, n_x, _, crps_x = _posterior_scores(model, data, t0, mc_itr, real, data_generator.expert_dim)
            mse_x.append((sse_x / n_x).cpu())        # (T', B): NaN where a patient has no observation at that step
            crps_x_all.append(crps_x.cpu().numpy())  # (T', B)
    mse_x = torch.cat(mse_x, dim=1)
    per_step = [_rmse_with_bootstrap(mse_x[i], resample_observed_only=False) for i in range(mse_x.shape[0])]
    cprs_x, cprs_x_sd = _mean_with_standard_error(np.concatenate(crps_x_all, axis=1), axis=1)
    return {"rmse_x": np.array([r for r, _ in per_step], dtype=np.float32), "rmse_x_sd": np.array([sd for _, sd in per_step]),
            "cprs_x": cprs_x, "cprs_x_sd": cprs_x_sd}
