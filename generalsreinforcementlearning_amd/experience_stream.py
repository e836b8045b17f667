"""VecExperienceStreamClient — the `ExperienceStreamClient` a trainer already uses, fed by a local engine instead of gRPC.

Reference: python/experience_stream_client.py - ExperienceConfig (:32-41), ExperienceStreamClient (:44-185: connect /
disconnect / start_streaming / stop_streaming / get_experience / get_batch / get_stats, a bounded queue filled by a
background thread, one dict per experience with the keys of `_process_experience`, drops counted when the queue is full),
ExperienceDataset (:188-218: fill_buffer / sample).  Same constructor arguments, same methods, same statistics keys, same
dict keys and types - what differs is where batches come from: the reference iterates
`stub.StreamExperienceBatches(request)`; here `connect()` opens a *source* - any iterator of batches (lists of experience
dicts) - and `engine_experience_source` is the one that matters: it plays a VecEngine with the on-device agent, takes the
compact experience records of a slice of the boards every step, expands them on the GPU
(gvec_expand_experience_records) and groups them the way StreamAggregator does (BatchProcessor: `batch_size` experiences
or `max_batch_wait_ms`, internal/grpc/gameserver/stream_aggregator.go:64-69), honouring the request's game / player
filters (experience_service.go filters by game id and player id).

Pinned by the reference's own client: tests/golden/make_stream_client_fixtures.py drives the reference's
ExperienceStreamClient / ExperienceDataset with hand-built ExperienceBatch messages (no server) and records queue
contents, drop counts, statistics, get_batch results and the dataset's draws; tests/test_experience_stream.py replays the
same batches through this class.
"""
import logging
import threading
import time
from dataclasses import dataclass
from queue import Empty, Queue
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional

import numpy as np

logger = logging.getLogger(__name__)


@dataclass
class ExperienceConfig:
    """experience_stream_client.py:32-41, field for field (server_address names the source in log lines only)."""
    server_address: str = "localhost:50051"
    game_ids: List[str] = None
    player_ids: List[int] = None
    batch_size: int = 32
    follow: bool = True
    enable_compression: bool = False
    max_batch_wait_ms: int = 100
    buffer_size: int = 1000


class VecExperienceStreamClient:
    """experience_stream_client.py:44-185 over a batch source.  `source_factory(config)` -> an iterator of batches, a batch
    being an iterable of experience dicts (keys of `_process_experience`); it is opened by connect() - the reference's
    channel + stub - and closed by disconnect() if it has a close()."""

    def __init__(self, config: ExperienceConfig, source_factory: Callable[[ExperienceConfig], Iterable]):
        self.config = config
        self.source_factory = source_factory
        self.source: Optional[Iterator] = None
        self.experience_queue = Queue(maxsize=config.buffer_size)
        self.streaming_thread = None
        self.stop_event = threading.Event()
        self.stats = {"total_experiences": 0, "total_batches": 0, "dropped_experiences": 0, "last_batch_time": None}

    def connect(self):
        self.source = iter(self.source_factory(self.config))
        logger.info("Connected to the experience source (%s)", self.config.server_address)

    def disconnect(self):
        src, self.source = self.source, None
        if src is not None and hasattr(src, "close"):
            src.close()
            logger.info("Disconnected from the experience source")

    def start_streaming(self):
        if self.streaming_thread and self.streaming_thread.is_alive():
            logger.warning("Streaming already started")
            return
        self.stop_event.clear()
        self.streaming_thread = threading.Thread(target=self._stream_worker)
        self.streaming_thread.daemon = True
        self.streaming_thread.start()
        logger.info("Started experience streaming")

    def stop_streaming(self):
        self.stop_event.set()
        if self.streaming_thread:
            self.streaming_thread.join(timeout=5)
        logger.info("Stopped experience streaming")

    def _stream_worker(self):
        try:
            for batch in self.source:
                if self.stop_event.is_set():
                    break
                self._process_batch(batch)
        except Exception as e:  # noqa: BLE001 - the reference logs and ends the stream (:111-114)
            logger.error("Unexpected error in streaming: %s", e)

    def _process_batch(self, batch):
        """:116-132: statistics, then every experience into the queue without blocking; a full queue drops."""
        self.stats["total_batches"] += 1
        self.stats["last_batch_time"] = time.time()
        for exp in batch:
            try:
                self.experience_queue.put(self._process_experience(exp), block=False)
                self.stats["total_experiences"] += 1
            except Exception:  # noqa: BLE001 - queue.Full, like the reference's bare except
                self.stats["dropped_experiences"] += 1

    @staticmethod
    def _process_experience(exp) -> Dict[str, Any]:
        """:134-158 for an experience that already is a dict of arrays: the same keys and types (float32 [9, H, W] tensors,
        bool mask or None, plain Python scalars)."""
        mask = exp.get("action_mask")
        return {"experience_id": exp["experience_id"], "game_id": exp["game_id"], "player_id": int(exp["player_id"]),
                "turn": int(exp["turn"]), "state": np.asarray(exp["state"], np.float32), "action": int(exp["action"]),
                "reward": float(exp["reward"]), "next_state": np.asarray(exp["next_state"], np.float32), "done": bool(exp["done"]),
                "action_mask": None if mask is None or len(mask) == 0 else np.asarray(mask, np.bool_)}

    def get_experience(self, timeout: float = 1.0) -> Optional[Dict[str, Any]]:
        try:
            return self.experience_queue.get(timeout=timeout)
        except Empty:
            return None

    def get_batch(self, batch_size: int, timeout: float = 5.0) -> List[Dict[str, Any]]:
        batch = []
        deadline = time.time() + timeout
        while len(batch) < batch_size and time.time() < deadline:
            exp = self.get_experience(timeout=0.1)
            if exp:
                batch.append(exp)
        return batch

    def get_stats(self) -> Dict[str, Any]:
        return {**self.stats, "queue_size": self.experience_queue.qsize(),
                "streaming": self.streaming_thread.is_alive() if self.streaming_thread else False}


class ExperienceDataset:
    """experience_stream_client.py:188-218: a buffer over the client, sampled without replacement with np.random."""

    def __init__(self, client, buffer_size: int = 10000):
        self.client = client
        self.buffer = []
        self.buffer_size = buffer_size

    def fill_buffer(self, min_size: int = 1000):
        while len(self.buffer) < min_size:
            batch = self.client.get_batch(100, timeout=1.0)
            if not batch:
                break
            self.buffer.extend(batch)
        if len(self.buffer) > self.buffer_size:
            self.buffer = self.buffer[-self.buffer_size:]

    def sample(self, batch_size: int) -> List[Dict[str, Any]]:
        if len(self.buffer) < batch_size:
            self.fill_buffer(batch_size)
        if len(self.buffer) < batch_size:
            return self.buffer.copy()
        indices = np.random.choice(len(self.buffer), batch_size, replace=False)
        return [self.buffer[i] for i in indices]


def engine_experience_source(engine, records_per_step, seed=0, max_steps=None, game_id_prefix="vec", invalid_permille=0):
    """-> source_factory for VecExperienceStreamClient: plays `engine` (a VecEngine with auto_reset and a board pool, its
    stream the current torch stream) with the on-device agent; every step the compact experience records of envs
    [0, records_per_step) are written, expanded on the GPU and cut into batches of config.batch_size (a partial batch is
    flushed when config.max_batch_wait_ms has passed: BatchProcessor).  config.game_ids / player_ids filter like the
    service does; config.follow = False ends the stream after max_steps steps (a bounded replay), True keeps going until
    the consumer stops."""
    import torch

    from .experience import ExperienceBatcher, RecordExpander

    def factory(config):
        lay = engine.experience_record_layout()
        slab = torch.empty(records_per_step * engine.experience_record_bytes(), dtype=torch.uint8, device="cuda")
        ex = RecordExpander(lay, records_per_step, slab.device)
        batcher = ExperienceBatcher(config.batch_size, config.max_batch_wait_ms / 1000.0)
        games = None if not config.game_ids else set(config.game_ids)
        players = None if not config.player_ids else set(int(p) for p in config.player_ids)
        engine.record_agent_actions(True)
        serial, step = 0, 0
        while max_steps is None or step < max_steps or config.follow:
            engine.experience_begin_range(0, records_per_step)
            engine.rollout(1, seed, invalid_permille, fused=False, want_stats=False)
            engine.experience_records(slab.data_ptr(), None, 0, records_per_step, 0)
            b = ex.compact(ex.expand(slab))
            step += 1
            host = {k: v.cpu().numpy() for k, v in b.items()}
            exps = []
            for i in range(len(host["env"])):
                gid, pid = f"{game_id_prefix}-env{int(host['env'][i])}", int(host["player_id"][i])
                if (games is not None and gid not in games) or (players is not None and pid not in players):
                    continue
                w, h = int(host["width"][i]), int(host["height"][i])
                serial += 1
                exps.append({"experience_id": f"{game_id_prefix}-{serial}", "game_id": gid, "player_id": pid, "turn": int(host["turn"][i]),
                             "state": host["state"][i][: 9 * w * h].reshape(9, h, w), "action": int(host["action"][i]),
                             "reward": float(host["reward"][i]), "next_state": host["next_state"][i][: 9 * w * h].reshape(9, h, w),
                             "done": bool(host["done"][i]), "action_mask": host["action_mask"][i][: 4 * w * h]})
            for full in batcher.add(exps) + batcher.poll():
                yield full
            if max_steps is not None and step >= max_steps and not config.follow:
                break
        last = batcher.flush()
        if last:
            yield last

    return factory
