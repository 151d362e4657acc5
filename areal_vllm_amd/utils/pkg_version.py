"""Installed-package version checks (``areal/utils/pkg_version.py:7-53``)."""
from importlib.metadata import PackageNotFoundError
from importlib.metadata import version as _version

from packaging.version import Version


def is_available(pkg_name: str) -> bool:
    try:
        return bool(_version(pkg_name))
    except PackageNotFoundError:
        return False


def compare_versions(version1: str, version2: str) -> int:
    a, b = Version(version1), Version(version2)
    return (a > b) - (a < b)


def is_version_greater_or_equal(package_name: str, target_version: str) -> bool:
    return compare_versions(_version(package_name), target_version) >= 0


def is_version_less(package_name: str, target_version: str) -> bool:
    return compare_versions(_version(package_name), target_version) < 0
