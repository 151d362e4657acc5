"""Installed-package version checks (the helpers of ``areal/utils/pkg_version.py:7-53``): everything is phrased in
terms of one lookup, `installed_version`, which answers None for a package that is not installed."""
from importlib import metadata
from typing import Optional

from packaging.version import InvalidVersion, Version


def installed_version(package_name: str) -> Optional[Version]:
    try:
        return Version(metadata.version(package_name))
    except (metadata.PackageNotFoundError, InvalidVersion):
        return None


def is_available(pkg_name: str) -> bool:
    return installed_version(pkg_name) is not None


def compare_versions(version1: str, version2: str) -> int:
    """-1 / 0 / 1 as version1 is older than / equal to / newer than version2 (PEP 440 ordering)."""
    left, right = Version(version1), Version(version2)
    if left == right:
        return 0
    return 1 if left > right else -1


def _require(package_name: str) -> Version:
    found = installed_version(package_name)
    if found is None:
        raise metadata.PackageNotFoundError(package_name)
    return found


def is_version_greater_or_equal(package_name: str, target_version: str) -> bool:
    return _require(package_name) >= Version(target_version)


def is_version_less(package_name: str, target_version: str) -> bool:
    return _require(package_name) < Version(target_version)
