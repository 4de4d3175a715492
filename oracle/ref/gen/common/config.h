// Stand-in for the cmake-generated common/config.h of the reference build (L/common/config.h.in): version and
// build-time strings only.  Fixed values, so that the VCF headers of the test binaries built from the reference's
// translation units (oracle/_ref/bin/*) do not depend on when they were built.  TEST INFRASTRUCTURE ONLY.
#pragma once
#define WORKFLOW_VERSION "2.9.x-oracle-build"
#define BUILD_TIME "1970-01-01T00:00:00Z"
#define CXX_COMPILER_NAME "g++"
#define COMPILER_VERSION "shim"
