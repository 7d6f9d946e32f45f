"""Drop-in for the reference's trainer.py: `from trainer import training`, same signature
(trainer.py:15) and epoch bookkeeping; the step body runs on the HIP path:

    forward            unet(images)                        -> libunet_hip unet_forward
    loss (L1)          BCEWithLogitsLoss(weight=class map)  -> unet_bce_step (target from labels, fwd + grad in one pass)
    backward           loss.backward()                     -> unet_backward_stage x6 (+ RCCL all-reduce if DP)
    update (L3)        SGD(lr=1e-4, momentum=0.99)          -> unet_sgd_momentum
    prediction (L2)    preds.argmax(dim=1) + IoU/PE        -> unet_eval_masks (fused crop+argmax+counts)
    weight map (N3)    class_balance(labels)               -> unet_class_balance

Reference behaviours kept on purpose: Q3 the dataset-name comparisons are IDENTITY tests
against string literals in the reference (trainer.py:18-27,68,110): a name arriving from argv never
passes them, a caller's literal 'ISBI2012' does (identifier-like constants are interned) and arms
the stop goal, and 'DIC-C2DH-HeLa' / 'PhC-C2DH-U373' never do (not interned), so class_balance is
always used (pinned by tests/golden/trainer_golden.json, made by running the reference);
Q4 the [B,H,W] weight map is right-aligned against [B,2,H,W] (works for B in {1,2}, raises
otherwise); Q5 only the first sample's metrics are kept per epoch.
"""
import os
import sys
from time import time

import numpy as np
import torch

from functions import class_balance, metrics_from_counts
import optim as hip_optim


def maybe_mkdir_p(path):
    os.makedirs(path, exist_ok=True)


_PROGRESS_FILES = (('train_iou', 'train_eval_iou.out'), ('train_pe', 'train_eval_pe.out'), ('val_iou', 'val_eval_iou.out'),
                   ('val_pe', 'val_eval_pe.out'), ('loss', 'loss.out'), ('loss_val', 'loss_val.out'))


def _f6(v):
    return "{:.6f}".format(v)


def _report(rows):
    """The epoch summary on stdout, label column 27 wide as in the reference's prints (trainer.py:150-160)."""
    for label, text in rows:
        print('%-27s %s' % (label, text))
    print(' ')


def _save_weights(unet, fold_dir, tag, before=None):
    """models/unet_weight_save_<tag>.pth + the reference's two announcement lines (trainer.py:143-147 and its siblings)."""
    path = os.path.join(fold_dir, 'models', 'unet_weight_save_{}.pth'.format(tag))
    torch.save(unet.state_dict(), path)
    if before is not None:
        print(before)
    print('Model has been saved:')
    print(path)


def _goal_for(DATASET):
    """(when_to_stop, goal) exactly as the reference's `DATASET is '<literal>'` chain decides it
    (trainer.py:18-27).  Which object a caller can hold: CPython interns identifier-like string
    constants, so any module's literal 'ISBI2012' IS the reference's constant (sys.intern returns
    that same object); 'DIC-C2DH-HeLa' and 'PhC-C2DH-U373' contain '-', are not interned, and the
    reference's private constant objects can be reached by no caller: those branches never fire.
    The ISBI2012 goal (a pixel-error value, 0.0611) is tested against the validation IoU, as there."""
    if DATASET is sys.intern('ISBI2012'):
        return 1, 0.0611
    return None, None


def _step_loss(unet, images, labels, device, train):
    preds = unet(images.to(device))
    labels = labels.to(device)                                  # everything below stays on the device
    pad = int((preds.shape[-1] - labels.shape[-1]) / 2)
    preds = preds[:, :, pad:labels.shape[-1] + pad, pad:labels.shape[-1] + pad]
    weight_maps = class_balance(labels.squeeze(1))              # [B,H,W] (trainer.py:72), unet_class_balance
    # the one-hot target [1-y, y] (trainer.py:63-66) is formed inside the kernel from the integer labels
    loss, _ = hip_optim.bce_argmax_step(preds, labels, weight=weight_maps, want_mask=False)
    return preds, loss, labels


def _first_sample_metrics(preds, labels):
    """argmax + IoU / pixel error of the batch in one device pass; only sample 0 is kept (quirk Q5)."""
    _, stats = hip_optim.crop_argmax_metrics(preds.detach(), labels)
    inter, union, diff = [int(v) for v in stats[0].tolist()]
    return metrics_from_counts(inter, union, diff, labels.shape[-1] * labels.shape[-2])


def training(unet, train_loader, val_loader, epochs, batch_size, device, fold_dir, DATASET, *, resume_from=None,
             save_optimizer=False):
    """Reference signature (trainer.py:15) plus two keyword-only extensions, both off by default so
    that the files written are exactly the reference's: resume_from = a checkpoint made by checkpoint.py (weights +
    SGD momentum + scheduler + epoch), save_optimizer = also write models/checkpoint_latest.pth every epoch."""
    when_to_stop, goal = _goal_for(DATASET)

    optimizer = hip_optim.SGD(unet.parameters(), lr=0.0001, momentum=0.99)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.1, patience=30,
                                                           threshold=1e-3, threshold_mode='rel', eps=1e-7)
    my_patience = 0
    first_epoch = 0
    if resume_from is not None:
        import checkpoint
        extra = checkpoint.load_checkpoint(resume_from, unet, optimizer, scheduler)
        first_epoch = int(extra.get("epoch", -1)) + 1
        my_patience = int(extra.get("my_patience", 0))
        resumed_best = extra.get("loss_best_epoch")
        if extra.get("goal_armed") is False:                   # the goal had been reached (and cleared) before the interruption
            when_to_stop = None
    else:
        resumed_best = None
        extra = {}

    maybe_mkdir_p(os.path.join(fold_dir, 'progress'))
    maybe_mkdir_p(os.path.join(fold_dir, 'models'))

    loss_best_epoch = 100000.0 if resumed_best is None else float(resumed_best)
    progress = {k: None for k in ('train_iou', 'train_pe', 'val_iou', 'val_pe', 'loss', 'loss_val')}
    for k, v in (extra.get("progress") or {}).items():         # a resumed run goes on writing the interrupted run's series
        if k in progress and len(v):
            progress[k] = np.array(v, dtype=np.float64)

    epoch = first_epoch - 1
    for epoch in range(first_epoch, epochs + 1):
        print(' ')
        print('Epoch:', epoch)
        start = time()
        total_loss = 0
        total_loss_val = 0
        train_eval = None
        val_eval = None

        for images, labels in train_loader:
            optimizer.zero_grad()
            preds, loss, labels = _step_loss(unet, images, labels, device, True)
            loss.backward()
            optimizer.step()
            total_loss += loss.detach()
            if train_eval is None:                              # Q5: first sample of the epoch only
                train_eval = _first_sample_metrics(preds, labels)
        train_eval_epoch = np.mean(train_eval, axis=1)

        with torch.no_grad():
            for images, labels in val_loader:
                preds, loss, labels = _step_loss(unet, images, labels, device, False)
                total_loss_val += loss
                if val_eval is None:
                    val_eval = _first_sample_metrics(preds, labels)
        val_eval_epoch = np.mean(val_eval, axis=1)

        scheduler.step(total_loss_val / (len(val_loader) * batch_size))
        for param_group in optimizer.param_groups:
            l_rate = param_group['lr']

        loss_epoch = total_loss / (len(train_loader) * batch_size)
        loss_epoch_val = total_loss_val / (len(val_loader) * batch_size)

        improved = loss_epoch_val < (loss_best_epoch * (1.0 - scheduler.threshold))
        if improved:
            loss_best_epoch = loss_epoch_val
            print('New best epoch!')
            _save_weights(unet, fold_dir, 'best')
        my_patience = 0 if improved else my_patience + 1

        _report((('Current lr is:', l_rate),
                 ('Patience is:', '{}/{}'.format(my_patience, scheduler.patience)),
                 ('Mean IoU training:', _f6(train_eval_epoch[0])), ('Mean PE training:', _f6(train_eval_epoch[1])),
                 ('Mean IoU validation:', _f6(val_eval_epoch[0])), ('Mean PE validation:', _f6(val_eval_epoch[1])),
                 ('Total training loss:', _f6(loss_epoch.item())), ('Total validation loss:', _f6(loss_epoch_val.item())),
                 ('Best epoch validation loss:', _f6(float(loss_best_epoch))),
                 ('Epoch duration:', _f6(time() - start) + ' s')))

        # the six progress series, one value per epoch, rewritten whole every epoch (trainer.py:162-183)
        new = dict(train_iou=train_eval_epoch[0], train_pe=train_eval_epoch[1], val_iou=val_eval_epoch[0],
                   val_pe=val_eval_epoch[1], loss=loss_epoch.item(), loss_val=loss_epoch_val.item())
        for k, fname in _PROGRESS_FILES:
            progress[k] = np.array([new[k]]) if progress[k] is None else np.append(progress[k], [new[k]])
            np.savetxt(os.path.join(fold_dir, 'progress', fname), progress[k])

        if save_optimizer:
            import checkpoint
            # state as it is at the END of this epoch: the patience reset and the goal test below are anticipated
            armed_after = when_to_stop is not None and not (val_eval_epoch[0] > goal)
            patience_after = my_patience
            if when_to_stop is None and my_patience == scheduler.patience:
                patience_after = -1
            checkpoint.save_checkpoint(os.path.join(fold_dir, 'models', 'checkpoint_latest.pth'), unet, optimizer, scheduler,
                                       epoch=epoch, my_patience=patience_after, loss_best_epoch=float(loss_best_epoch),
                                       goal_armed=armed_after, progress={k: [float(x) for x in v] for k, v in progress.items()})

        if when_to_stop is not None:
            # goal armed (trainer.py:185-214): the periodic checkpoint and the LR-floor stop are skipped this epoch
            if val_eval_epoch[0] > goal:
                _save_weights(unet, fold_dir, DATASET, before='The goal was reached in epoch {}!'.format(epoch))
                when_to_stop = None
            continue

        if epoch % 25 == 0:
            _save_weights(unet, fold_dir, 'latest')

        lr_floor = 10 * scheduler.eps
        out_of_patience = my_patience == scheduler.patience
        if l_rate < lr_floor and out_of_patience:
            for line in (f'LR dropped below {lr_floor}!', 'Stopping training', ' '):
                print(line)
            _save_weights(unet, fold_dir, 'latest')
            break
        if out_of_patience:
            my_patience = -1

    print('Training is finished as epoch {} has been reached'.format(epoch))
    print(' ')
