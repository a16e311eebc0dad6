"""`Trainer`: the epoch loop around `step_feedfwd`, API of /root/reference/common/train.py:55-320.

Host code only: it reads the same .ini sections and keys as the reference, builds the
DataLoaders, alternates validation / checkpoint / lr adjustment / training epochs in the reference's order
and writes the reference's checkpoint dict (`epoch_NNN.pth.tar`).  Every step is one `step_feedfwd` call,
i.e. one fused library call on the HIP path.

Not carried over: visdom plotting (a `[logging] visdom = yes` config is accepted and ignored with a
notice), the LSTM / VidLoc step.  New: batches are prefetched to the device one step ahead (feed.py; MN_PREFETCH=0: the
reference's in-step copy); under torch.distributed (one process per GPU) the training
windows are sharded with a DistributedSampler and rank 0 alone prints and writes checkpoints.
"""
import configparser
import os
import os.path as osp
import time

import torch
import torch.utils.data
from torch.utils.data.dataloader import default_collate

from . import dp
from .feed import DeviceFeed
from .posenet import engine_of
from .train import load_checkpoint, save_checkpoint, step_feedfwd


class AverageMeter:
    """running value / average (common/Logger.py's meter as the loop uses it)"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def safe_collate(batch):
    """default_collate over the samples that are not None (common/train.py:55-62)"""
    return default_collate([b for b in batch if b is not None])


def read_config(config_file):
    """the [training] / [logging] keys Trainer.__init__ reads (common/train.py:93-112); `config_file` is a path
    or an already parsed ConfigParser"""
    if isinstance(config_file, configparser.ConfigParser):
        settings = config_file
    else:
        settings = configparser.ConfigParser()
        with open(config_file, "r") as f:
            settings.read_file(f)
    config = {}
    section = settings["training"]
    config["n_epochs"] = section.getint("n_epochs")
    config["batch_size"] = section.getint("batch_size")
    config["do_val"] = section.getboolean("do_val")
    config["shuffle"] = section.getboolean("shuffle")
    config["seed"] = section.getint("seed")
    config["num_workers"] = section.getint("num_workers")
    config["snapshot"] = section.getint("snapshot")
    config["val_freq"] = section.getint("val_freq")
    config["max_grad_norm"] = section.getfloat("max_grad_norm", 0)
    section = settings["logging"]
    config["log_visdom"] = section.getboolean("visdom")
    config["print_freq"] = section.getint("print_freq")
    return config


class Trainer:
    def __init__(self, model, optimizer, train_criterion, config_file, experiment, train_dataset, val_dataset, device,
                 checkpoint_file=None, resume_optim=False, val_criterion=None, logdir=None, log=print):
        self.model = model
        self.train_criterion = train_criterion
        self.val_criterion = train_criterion if val_criterion is None else val_criterion
        self.experiment = experiment
        self.optimizer = optimizer
        if "CUDA_VISIBLE_DEVICES" not in os.environ and device is not None and dp.world_size() == 1:
            os.environ["CUDA_VISIBLE_DEVICES"] = device

        self.config = read_config(config_file)
        self.config["cuda"] = torch.cuda.is_available()
        self.rank = torch.distributed.get_rank() if dp.world_size() > 1 else 0
        self._log = log if self.rank == 0 else (lambda *a, **k: None)
        if self.config["log_visdom"]:
            self._log("visdom logging is not part of the MI355X hot path: ignored")

        self.logdir = logdir if logdir is not None else osp.join(os.getcwd(), "logs", self.experiment)
        if self.rank == 0:
            os.makedirs(self.logdir, exist_ok=True)

        self._log("---------------------------------------")
        self._log("Experiment: {:s}".format(self.experiment))
        for k, v in self.config.items():
            self._log("{:s}: {:s}".format(k, str(v)))
        self._log("---------------------------------------")

        torch.manual_seed(self.config["seed"])

        self.start_epoch = 0
        if checkpoint_file:
            if osp.isfile(checkpoint_file):
                self.start_epoch = load_checkpoint(checkpoint_file, self.model, self.optimizer, self.train_criterion,
                                                   resume_optim=resume_optim)
                self._log("Loaded checkpoint {:s} epoch {:d}".format(checkpoint_file, self.start_epoch))

        pin = self.config["cuda"]
        self.train_sampler = None
        if dp.world_size() > 1:
            self.train_sampler = torch.utils.data.distributed.DistributedSampler(
                train_dataset, shuffle=self.config["shuffle"], seed=self.config["seed"], drop_last=True)
        self.train_loader = torch.utils.data.DataLoader(
            train_dataset, batch_size=self.config["batch_size"],
            shuffle=self.config["shuffle"] and self.train_sampler is None, sampler=self.train_sampler,
            num_workers=self.config["num_workers"], pin_memory=pin, collate_fn=safe_collate,
            drop_last=dp.world_size() > 1)  # equal local batches: the gradient mean over ranks needs them
        if self.config["do_val"]:
            self.val_loader = torch.utils.data.DataLoader(
                val_dataset, batch_size=self.config["batch_size"], shuffle=self.config["shuffle"],
                num_workers=self.config["num_workers"], pin_memory=pin, collate_fn=safe_collate)
        else:
            self.val_loader = None

        if self.config["cuda"]:
            self.model.cuda()
            self.train_criterion.cuda()
            self.val_criterion.cuda()
        # the reference copies each pinned batch with .cuda(async=True) inside step_feedfwd (common/train.py:341,347), in front of
        # the step on its own stream; here the copy of batch k+1 is issued on a copy stream while step k runs (feed.py)
        self.prefetch = self.config["cuda"] and os.environ.get("MN_PREFETCH", "1") != "0"

    def _feed(self, loader):
        return DeviceFeed(loader, engine_of(self.model).device) if self.prefetch else loader

    def save_checkpoint(self, epoch):
        filename = osp.join(self.logdir, "epoch_{:03d}.pth.tar".format(epoch))
        if self.rank == 0:
            save_checkpoint(filename, epoch, self.model, self.optimizer, self.train_criterion)
        return filename

    def train_val(self, lstm=False):
        """the reference's loop (common/train.py:206-320); returns the last training loss"""
        if lstm:
            raise NotImplementedError("the LSTM (VidLoc) step is outside the MapNet hot path")
        cfg = self.config
        loss = float("nan")
        for epoch in range(self.start_epoch, cfg["n_epochs"]):
            # VALIDATION
            if cfg["do_val"] and ((epoch % cfg["val_freq"] == 0) or (epoch == cfg["n_epochs"] - 1)):
                val_batch_time, val_data_time, val_loss = AverageMeter(), AverageMeter(), AverageMeter()
                self.model.eval()
                end = time.time()
                for batch_idx, (data, target) in enumerate(self._feed(self.val_loader)):
                    val_data_time.update(time.time() - end)
                    vloss, _ = step_feedfwd(data, self.model, cfg["cuda"], target=target, criterion=self.val_criterion,
                                            optim=self.optimizer, train=False)
                    val_loss.update(vloss)
                    val_batch_time.update(time.time() - end)
                    if batch_idx % cfg["print_freq"] == 0:
                        self._log("Val {:s}: Epoch {:d}\tBatch {:d}/{:d}\tData time {:.4f} ({:.4f})\t"
                                  "Batch time {:.4f} ({:.4f})\tLoss {:f}".format(
                                      self.experiment, epoch, batch_idx, len(self.val_loader) - 1, val_data_time.val,
                                      val_data_time.avg, val_batch_time.val, val_batch_time.avg, vloss))
                    end = time.time()
                self._log("Val {:s}: Epoch {:d}, val_loss {:f}".format(self.experiment, epoch, val_loss.avg))
                self.last_val_loss = val_loss.avg

            # SAVE CHECKPOINT
            if epoch % cfg["snapshot"] == 0:
                self.save_checkpoint(epoch)
                self._log("Epoch {:d} checkpoint saved for {:s}".format(epoch, self.experiment))

            # ADJUST LR
            lr = self.optimizer.adjust_lr(epoch)

            # TRAIN
            self.model.train()
            if self.train_sampler is not None:
                self.train_sampler.set_epoch(epoch)
            train_data_time, train_batch_time = AverageMeter(), AverageMeter()
            end = time.time()
            for batch_idx, (data, target) in enumerate(self._feed(self.train_loader)):
                train_data_time.update(time.time() - end)
                loss, _ = step_feedfwd(data, self.model, cfg["cuda"], target=target, criterion=self.train_criterion,
                                       optim=self.optimizer, train=True, max_grad_norm=cfg["max_grad_norm"])
                train_batch_time.update(time.time() - end)
                if batch_idx % cfg["print_freq"] == 0:
                    self._log("Train {:s}: Epoch {:d}\tBatch {:d}/{:d}\tData Time {:.4f} ({:.4f})\t"
                              "Batch Time {:.4f} ({:.4f})\tLoss {:f}\tlr: {:f}".format(
                                  self.experiment, epoch, batch_idx, len(self.train_loader) - 1, train_data_time.val,
                                  train_data_time.avg, train_batch_time.val, train_batch_time.avg, loss, lr))
                end = time.time()

        # Save final checkpoint
        epoch = cfg["n_epochs"]
        self.final_checkpoint = self.save_checkpoint(epoch)
        self._log("Epoch {:d} checkpoint saved".format(epoch))
        return loss
