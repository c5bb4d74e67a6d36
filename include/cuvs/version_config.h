/* Library version — the header rapids-cmake generates for the reference (cpp/CMakeLists.txt:36,
 * rapids_cmake_write_version_file); checked against cuvsVersionGet by c/tests/core/c_api.c:80-86. */
#pragma once
#define CUVS_VERSION_MAJOR 26
#define CUVS_VERSION_MINOR 8
#define CUVS_VERSION_PATCH 0
