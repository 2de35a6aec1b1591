function [tform, inlierIdx, isFound] = estimateTransformationMLESAC(points1, points2, transformationType, input)
    %ESTIMATETRANSFORMATIONMLESAC Shadows PP/imageMatching/estimateTransformationMLESAC.m ('projective').
    %   The 4-point draws are generated here with randperm exactly as the reference does (:160) and handed to
    %   the device as an explicit input; fitting, truncated-loss scoring, the adaptive stop and the refit on the
    %   inliers run in aps_mex.
    if nargin < 4, input = struct(); end
    if ~strcmpi(transformationType, 'projective')
        % MLESAC's estimators of the other model classes (:389-640) run the reference's own host code; with
        % imageMatchingMethod 'ransac' all five types run on the device
        [tform, inlierIdx, isFound] = aps_call_shadowed('estimateTransformationMLESAC', mfilename('fullpath'), ...
            points1, points2, transformationType, input);
        return;
    end
    if size(points1, 1) ~= size(points2, 1)
        error('estimateTransformationMLESAC:PointCountMismatch', 'points1 and points2 must have the same number of rows.');
    end
    M = size(points1, 1);
    if M < 4
        tform = []; inlierIdx = false(M, 1); isFound = false; return;
    end
    if isfield(input, 'maxIter'), S = input.maxIter + 64; else, S = 1064; end
    sampleIdx = zeros(4, S, 'uint32');
    for s = 1:S, sampleIdx(:, s) = uint32(randperm(M, 4)); end
    [tform, inlierIdx, isFound] = aps_mex('mlesac_homography', double(points1), double(points2), input, sampleIdx);
    if ~isFound, tform = []; end
end
