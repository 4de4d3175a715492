#pragma once
#include <cerrno>
