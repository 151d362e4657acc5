"""Batched HTTP client of the remote function-call (sandbox) service.

``batch_function_call(payloads, task_type, timeout)`` posts every payload to
``$FUNCTIONCALL_SERVICE_DOMAIN/apis/functioncalls`` with bounded concurrency, retries with exponential
back-off + jitter, maps time-outs and exhausted retries to a failed result instead of raising, treats a
``SystemError`` result as retryable and logs p50 / p90 / p99 latencies. Behaviour of
``functioncall/base/call.py:81-235`` in the reference.
"""
import asyncio
import enum
import os
import random
import time
import uuid
from statistics import median
from typing import Any, Dict, List, Optional, Tuple

import aiohttp

from ..utils import logging

logger = logging.getLogger("functioncall")
FUNCTIONCALL_SERVICE_DOMAIN = os.getenv("FUNCTIONCALL_SERVICE_DOMAIN", "")


class Language(enum.Enum):
    PYTHON = 0
    JAVA = 1
    CPP = 2
    C = 3
    MATH = 4
    SQL = 5
    GO = 6
    NODEJS = 7
    CSHARP = 8
    TYPESCRIPT = 9
    JAVASCRIPT = 10

    def __str__(self):
        return self.name.lower()


def construct_uid(query_id: str, start_idx: int, end_idx: int) -> str:
    return f"[{uuid.uuid4()}]-{query_id}-[{start_idx}-{end_idx}]"


def calculate_percentile(elapsed_times: List[float], percentile: float) -> float:
    s = sorted(elapsed_times)
    return s[min(int(len(s) * percentile / 100), len(s) - 1)]


def has_system_error(response_json: Dict) -> Tuple[bool, Optional[Dict]]:
    for r in response_json.get("results", []):
        if r.get("errorType", "") == "SystemError":
            return True, r
    return False, None


def check_payload(payload: Optional[Dict]) -> Tuple[bool, Dict]:
    if not payload:
        return False, dict(uid="", success=False, results=[dict(success=False, reason="Empty payload",
                                                               errorType="UnknownError")])
    if not payload.get("code"):
        return False, dict(uid=payload.get("uid", ""), success=False,
                           results=[dict(success=False, reason="Empty code", errorType="UnknownError")])
    return True, {}


def _failed(uid: str, reason: str) -> Dict:
    return dict(uid=uid, success=False, results=[dict(success=False, reason=reason, errorType="UnknownError")])


async def async_invoke_function(session: aiohttp.ClientSession, url: str, timeout: aiohttp.ClientTimeout,
                                payload: Dict[str, Any], max_retries: int = 2, initial_retry_interval: float = 0.5,
                                max_retry_interval: float = 10.0) -> Dict:
    uid = payload.get("uid", "")
    for attempt in range(1, max_retries + 1):
        try:
            async with session.post(url, json=payload, timeout=timeout) as resp:
                if resp.status != 200:
                    raise RuntimeError(f"HTTP Error {resp.status}: {await resp.text()}")
                body = await resp.json()
                bad, info = has_system_error(body)
                if bad:
                    raise RuntimeError(f"SystemError detected, uid: {body.get('uid')}, err: {info}")
                return body
        except asyncio.TimeoutError:
            logger.warning(f"Request timeout after {timeout}s, uid: {uid}, URL: {url}")
            return _failed(uid, "Function call timed out.")
        except Exception as e:
            logger.error(f"Async invocation failed on attempt {attempt}: {e}, uid: {uid}, URL: {url}")
        if attempt >= max_retries:
            break
        await asyncio.sleep(min(initial_retry_interval * (2 ** attempt) + random.uniform(0, 5), max_retry_interval))
    return _failed(uid, "Function call exceed max retries.")


async def batch_function_call_async(payload_list: List[Dict], url: str, timeout, concurrency: int = 1500) -> List[Dict]:
    if not isinstance(timeout, aiohttp.ClientTimeout):
        timeout = aiohttp.ClientTimeout(total=timeout)
    connector = aiohttp.TCPConnector(limit=concurrency, ttl_dns_cache=300, keepalive_timeout=80)
    async with aiohttp.ClientSession(connector=connector) as session:
        sem = asyncio.Semaphore(concurrency)

        async def one(payload):
            ok, err = check_payload(payload)
            if not ok:
                return err, 0.0
            async with sem:
                t0 = time.monotonic()
                return await async_invoke_function(session, url, timeout, payload), time.monotonic() - t0

        results = await asyncio.gather(*[one(p) for p in payload_list])
    data = [r for r, _ in results]
    times = [t for _, t in results]
    if times:
        worst = max(range(len(times)), key=lambda i: times[i])
        logger.info(f"Longest functioncall took {times[worst]:.4f} seconds, uid: {data[worst].get('uid')}, p50: "
                    f"{median(times):.4f}, p90: {calculate_percentile(times, 90):.4f}, p99: "
                    f"{calculate_percentile(times, 99):.4f}")
    return data


def get_runtime_name(runtime: Optional[str], language) -> str:
    return runtime if runtime else str(language).lower() + "-default"


def calculate_concurrency(dp: Optional[int] = None, total: int = 5000) -> int:
    """Share a service-wide budget of concurrent sandbox cores between the data-parallel verifiers."""
    return max(1, total // (dp or 16))


def batch_function_call(payload_list: List[Dict], task_type: str, timeout, url: Optional[str] = None,
                        concurrency: Optional[int] = None) -> List[Dict]:
    t0 = time.time()
    url = url or f"{os.getenv('FUNCTIONCALL_SERVICE_DOMAIN', FUNCTIONCALL_SERVICE_DOMAIN)}/apis/functioncalls"
    concurrency = concurrency or calculate_concurrency()
    logger.info(f"Batch function call start, task type: {task_type}, request count: {len(payload_list)}, "
                f"concurrency: {concurrency}")
    out = asyncio.run(batch_function_call_async(payload_list, url, timeout, concurrency))
    logger.info(f"Batch function call done, task type: {task_type}, batch size: {len(payload_list)}, cost: "
                f"{(time.time() - t0) * 1000:.0f} ms")
    return out
